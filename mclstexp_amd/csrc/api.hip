#include "common.h"

extern "C" int mcl_abi_version(void) { return MCL_ABI_VERSION; }

extern "C" const char* mcl_error_string(int code) {
  switch (code) {
    case MCL_OK: return "ok";
    case MCL_EINVAL: return "invalid argument (null pointer, non-positive size or inconsistent layout)";
    case MCL_EUNSUPPORTED: return "unsupported configuration";
    case MCL_EWORKSPACE: return "workspace too small";
    default: return code > 0 ? hipGetErrorString((hipError_t)code) : "unknown mclstexp error";
  }
}
