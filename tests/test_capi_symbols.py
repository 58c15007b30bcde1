"""CPU: the C-ABI library loads and exports every symbol include/mclstexp_hip.h declares (no compute)."""
import os
import re

import pytest

from conftest import ROOT


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "mclstexp_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mcl_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_expected_surface():
    syms = declared_symbols()
    for s in ("mcl_gemm", "mcl_layernorm_fwd", "mcl_layernorm_bwd", "mcl_pos_embed_add_fwd", "mcl_infonce_lse",
              "mcl_infonce_dlogits", "mcl_adam_step", "mcl_adam_table_step"):
        assert s in syms


def test_library_loads_and_exports_every_declared_symbol():
    from mclstexp_amd import _lib, build
    if not os.path.exists(_lib.LIB_PATH):
        build.build(verbose=False)
    lib = _lib.load()
    assert lib.mcl_abi_version() == _lib.ABI_VERSION
    for s in declared_symbols():
        assert hasattr(lib, s), f"{s} declared in include/mclstexp_hip.h but not exported"
        assert s in _lib.PROTOTYPES, f"{s} has no ctypes prototype in mclstexp_amd/_lib.py"
    assert set(_lib.PROTOTYPES) == set(declared_symbols())
    assert b"invalid" in lib.mcl_error_string(-1)


def test_missing_library_fails_loudly(tmp_path):
    from mclstexp_amd import _lib
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _lib.load(str(tmp_path / "nope.so"))


def test_argument_errors_without_gpu():
    """Argument validation happens before any launch, so it is checkable on CPU."""
    import ctypes as C
    from mclstexp_amd import _lib
    lib = _lib.load()
    a = _lib.GemmArgs()
    assert lib.mcl_gemm(C.byref(a), None) == -1
    assert lib.mcl_layernorm_fwd(None, 0, None, None, None, 0, None, None, 0, 0, 1e-5, None) == -1
    assert lib.mcl_adam_step(None, None, None, None, 0, 1e-4, .9, .999, 1e-8, 1e-3, .1, .001, None) == -1
