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
    a = _lib.gemm_args()
    assert lib.mcl_gemm(C.byref(a), None) == -1
    assert lib.mcl_layernorm_fwd(None, 0, None, None, None, 0, None, None, 0, 0, 1e-5, None) == -1
    assert lib.mcl_adam_step(None, None, None, None, 0, 1e-4, .9, .999, 1e-8, 1e-3, .1, .001, None) == -1


# ---------------------------------------------------------------- struct layouts: header == ctypes binding == INTEGRATION.md
_CTYPES_OF = {"int32_t": "c_int", "uint32_t": "c_uint", "int64_t": "c_long", "float": "c_float", "double": "c_double"}


def header_structs():
    """{name: [(field, kind)]} for every ``typedef struct`` in include/mclstexp_hip.h; kind = a C scalar type or 'ptr'."""
    text = open(os.path.join(ROOT, "include", "mclstexp_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    out = {}
    for name, body in re.findall(r"typedef\s+struct\s+(\w+)\s*\{(.*?)\}\s*\w+\s*;", text, flags=re.S):
        fields = []
        for decl in body.split(";"):
            decl = " ".join(decl.split())
            if not decl:
                continue
            m = re.match(r"(const\s+)?(\w+)\s*(\*?)\s*(.*)", decl)
            ctype, star, names = m.group(2), m.group(3), m.group(4)
            for n in names.split(","):
                n = n.strip()
                ptr = bool(star) or n.startswith("*")
                fields.append((n.lstrip("* "), "ptr" if ptr else ctype))
        out[name] = fields
    return out


def _ctypes_fields(struct):
    import ctypes as C
    res = []
    for n, t in struct._fields_:
        if t is C.c_void_p or (isinstance(t, type) and issubclass(t, C._Pointer)):
            res.append((n, "ptr"))
        else:
            res.append((n, t.__name__))
    return res


def _snippet_struct(path, class_name):
    """Executes the ``class <class_name>(C.Structure)`` block of a markdown document's python snippet; returns the class."""
    text = open(path).read()
    m = re.search(r"^(class " + class_name + r"\(C\.Structure\):.*?)\n\n", text, flags=re.S | re.M)
    assert m, f"{path} holds no ctypes declaration of {class_name}"
    import ctypes as C
    ns = {"C": C, "vp": C.c_void_p, "i32": C.c_int32, "u32": C.c_uint32, "i64": C.c_int64, "f32": C.c_float}
    exec(m.group(1), ns)
    return ns[class_name]


def test_every_header_struct_matches_its_ctypes_declaration():
    import ctypes as C
    from mclstexp_amd import _lib
    bound = {"mcl_gemm_args": _lib.GemmArgs}
    structs = header_structs()
    assert set(structs) == set(bound), "a struct of include/mclstexp_hip.h has no ctypes declaration in mclstexp_amd/_lib.py"
    for name, fields in structs.items():
        want = [(n, "ptr" if k == "ptr" else _CTYPES_OF[k]) for n, k in fields]
        assert _ctypes_fields(bound[name]) == want, name
    lib = _lib.load()
    assert lib.mcl_gemm_args_size() == C.sizeof(_lib.GemmArgs)
    assert lib.mcl_gemm_args_min_size() == _lib.GemmArgs.workspace.offset + C.sizeof(C.c_void_p)


def test_integration_md_struct_is_the_current_layout():
    """INTEGRATION.md section 2's binding, executed: same fields, order and size as the library's struct."""
    import ctypes as C
    from mclstexp_amd import _lib
    doc = _snippet_struct(os.path.join(ROOT, "INTEGRATION.md"), "GemmArgs")
    assert _ctypes_fields(doc) == _ctypes_fields(_lib.GemmArgs)
    assert C.sizeof(doc) == _lib.load().mcl_gemm_args_size()


def test_gemm_args_struct_size_is_honoured():
    """mcl_gemm reads nothing beyond the caller's struct_size: a struct that ends at `workspace` (the ABI-6 fields) passes the
    argument check with garbage behind it, a shorter or unsized one is MCL_EINVAL.  Argument checks only (no GPU)."""
    import ctypes as C
    from mclstexp_amd import _lib
    lib = _lib.load()
    full = _lib.GemmArgs
    short_fields = full._fields_[:[n for n, _ in full._fields_].index("workspace") + 1]

    class Short(C.Structure):
        _fields_ = short_fields

    class Padded(C.Structure):                      # the short struct followed by bytes a stale binding never initialised
        _fields_ = [("a", Short), ("junk", C.c_ubyte * 64)]

    assert C.sizeof(Short) == lib.mcl_gemm_args_min_size()
    fn = lib.mcl_gemm
    fn_short = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p)(("mcl_gemm", lib))
    p = Padded()
    C.memset(C.byref(p, C.sizeof(Short)), 0xAB, 64)                    # flt_thr & co would read as wild pointers
    a = p.a
    a.M, a.N, a.K, a.batch = 4, 4, 4, 1
    a.A = a.B = a.C = 0x1000                                            # never dereferenced: the checks below fail first
    a.sAm, a.sAk, a.sBk, a.sBn, a.ldc = 4, 1, 1, 4, 4
    a.compute = 7                                                       # -> MCL_EUNSUPPORTED, after every pointer / filter check
    a.struct_size = C.sizeof(Short)
    assert fn_short(C.addressof(p), None) == -2, "fields beyond struct_size must read as zero (the filter block is not engaged)"
    a.struct_size = C.sizeof(Short) + 64                                # the caller now CLAIMS the junk: filter block engaged -> EINVAL
    assert fn_short(C.addressof(p), None) == -1
    a.struct_size = C.sizeof(Short) - 8
    assert fn_short(C.addressof(p), None) == -1
    a.struct_size = 0
    assert fn_short(C.addressof(p), None) == -1
    g = _lib.gemm_args()
    assert g.struct_size == C.sizeof(full)
    assert fn(C.byref(g), None) == -1                                   # sized, but null operands
