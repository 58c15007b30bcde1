// Shared device helpers for the gfx950 kernels (wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "../../include/mclstexp_hip.h"

#define MCL_WAVE 64

// hipGetLastError() is sticky per host thread: a benign failure inside the caller's own runtime use
// (PyTorch probes) would otherwise be reported as ours.  Clear it before enqueueing.
#define MCL_CLEAR_ERROR() (void)hipGetLastError()

#define MCL_CHECK_LAUNCH()                                  \
  do {                                                      \
    hipError_t e__ = hipGetLastError();                     \
    if (e__ != hipSuccess) return (int)e__;                 \
  } while (0)

static inline hipStream_t mcl_stream(mcl_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

// One-time PER-DEVICE setup (hipFuncSetAttribute(MaxDynamicSharedMemorySize) is a per-device attribute: a process-wide flag
// would leave a second GPU of the same process without it), safe against the autograd worker thread racing the main thread.
#include <atomic>
struct mcl_device_once {
  std::atomic<unsigned long long> done{0};
  struct guard {
    mcl_device_once* o;
    unsigned long long bit;
    bool need;
    explicit operator bool() const { return need; }
    ~guard() {
      if (need) o->done.fetch_or(bit, std::memory_order_release);    // set AFTER the guarded block has run
    }
  };
  // usage: if (auto g = flag.first()) { ...hipFuncSetAttribute... }   -- two racing threads may both run the (idempotent)
  // block; none launches before the attribute of its device is set
  guard first() {
    int d = 0;
    (void)hipGetDevice(&d);
    const unsigned long long bit = 1ull << (d & 63);
    return guard{this, bit, !(done.load(std::memory_order_acquire) & bit)};
  }
};
// environment overrides of grid sizes: never below 1 (atoi of garbage / "0" / a negative value would launch a zero-size grid)
// compute units of the current device (cached per device id; 256 on MI355X): the grid of persistent kernels
static inline int mcl_cu_count() {
  static int cached[64];
  int d = 0;
  (void)hipGetDevice(&d);
  if (d < 0 || d >= 64) d = 0;
  if (cached[d] == 0) {
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, d) != hipSuccess || n <= 0) n = 256;
    cached[d] = n;
  }
  return cached[d];
}

static inline int mcl_env_grid(const char* value, int dflt) {
  if (!value) return dflt;
  const int v = atoi(value);
  return v < 1 ? 1 : v;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// exact-erf GELU (nn.GELU() default) and its derivative
__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float gelu_erf_grad(float x) {
  const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
  const float pdf = 0.39894228040143267794f * __expf(-0.5f * x * x);
  return cdf + x * pdf;
}

// csrc/wrw_fused.hip: dW[e] (+)= sum_{z < ks} wpart[z*MN + e] in fixed order (deterministic merge of per-workgroup
// weight-gradient partials); MN a multiple of 4, all pointers 16-byte aligned.  Enqueue only.
void mcl_launch_wrw_merge(const float* wpart, int ks, long long MN, float* dW, int accumulate_w, hipStream_t st);
// same with an explicit slab stride (floats)
void mcl_launch_wrw_merge_strided(const float* wpart, int ks, long long MN, long long stride, float* dW, int accumulate_w,
                                  hipStream_t st);

// csrc/conv3x3_rows.hip: row-walking form of the growth 3x3 convolution for the large maps (enqueue only).
bool mcl_conv3x3_rows_applicable(long long S, int H, int W);
long long mcl_conv3x3_rows_workspace_floats(long long S);
int mcl_launch_conv3x3_fwd_rows(const void* z, long long S, int H, int W, const float* gamma, const float* beta,
                                const float* mean, const float* rstd, const void* W2, void* out, long long ldo,
                                float* workspace, float eps, float* ymean, float* yvar, float* yrstd, hipStream_t st);

// csrc/conv3x3_wrw_rows.hip: row-walking form of the growth 3x3 weight gradient for the large maps (enqueue only;
// workspace = one fp32 partial per workgroup; launches the fixed-order merge itself).
bool mcl_conv3x3_wrw_rows_applicable(long long S, int H, int W);
long long mcl_conv3x3_wrw_rows_workspace_floats(long long S, int H, int W);
int mcl_launch_conv3x3_wrw_rows(const void* dy, long long lddy, const void* z, long long S, int H, int W, const float* gamma,
                                const float* beta, const float* mean, const float* rstd, float* workspace, float* dW,
                                int accumulate_w, hipStream_t st);
