"""The C-ABI library must load on a machine without a GPU and export every symbol include/gsr_hip.h declares,
with the argument lists the ctypes binding assumes.  No compute entry point is called here (CPU-only suite)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _protos():
    src = open(os.path.join(ROOT, "include", "gsr_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return re.findall(r"(?:int|size_t|const char\*)\s+(gsr_\w+)\s*\(([^;]*?)\)\s*;", src, flags=re.S)


def test_library_loads_and_exports_every_declared_symbol(hip_lib_built):
    import _gsr
    raw = ctypes.CDLL(_gsr.LIB_PATH)
    names = [n for n, _ in _protos()]
    assert len(names) >= 14
    for n in names:
        assert hasattr(raw, n), f"{n} declared in gsr_hip.h but not exported by libgsr_hip.so"
    assert sorted(names) == sorted(_gsr.EXPORTED)
    # the library and the binding agree on the ABI version the header states (a binding written against another header refuses to load)
    hdr = open(os.path.join(ROOT, "include", "gsr_hip.h")).read()
    assert _gsr.lib.gsr_version() == _gsr.GSR_ABI_VERSION == int(re.search(r"#define GSR_ABI_VERSION (\d+)", hdr).group(1))


def test_ctypes_signatures_match_header(hip_lib_built):
    import _gsr

    def ctype(arg):
        arg = arg.strip()
        if arg == "void":
            return None
        if arg.startswith("gsr_alloc_fn"):
            return _gsr.ALLOC_FN
        if arg.startswith("const gsr_adam_segment*"):
            return ctypes.POINTER(_gsr.AdamSegment)
        if arg.startswith("const gsr_gather_group*"):
            return ctypes.POINTER(_gsr.GatherGroup)
        if arg.startswith("const gsr_refl_forward*"):
            return ctypes.POINTER(_gsr.ReflForward)
        if "*" in arg:
            return ctypes.c_char_p if (arg.startswith("const char") and "uint8" not in arg) else ctypes.c_void_p
        for pre, t in (("float", ctypes.c_float), ("uint32_t", ctypes.c_uint32), ("uint64_t", ctypes.c_uint64), ("size_t", ctypes.c_size_t),
                       ("int", ctypes.c_int)):
            if arg.startswith(pre):
                return t
        raise ValueError(arg)

    for name, args in _protos():
        want = [c for c in (ctype(a) for a in args.split(",")) if c is not None]
        have = list(getattr(_gsr.lib, name).argtypes or [])
        assert want == have, name


def test_descriptor_structs_match_header(hip_lib_built):
    """The descriptor struct of the fused rasterize + reflect forward: same fields in the same order as the header."""
    import _gsr
    src = open(os.path.join(ROOT, "include", "gsr_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    for cname, cls in (("gsr_refl_forward", _gsr.ReflForward),):
        body = re.search(r"typedef struct \{([^}]*)\}\s*%s;" % cname, src, flags=re.S).group(1)
        fields = [re.sub(r".*[\s\*]", "", f.strip()) for f in body.split(";") if f.strip()]
        assert fields == [f[0] for f in cls._fields_], cname


def test_host_only_entry_points(hip_lib_built):
    """Calls that never touch the device: option switches, error strings, argument validation."""
    import _gsr
    assert _gsr.lib.gsr_set_option(b"cull", 1) == 0
    assert _gsr.lib.gsr_set_option(b"no-such-option", 1) == -1
    assert b"no-such-option" in _gsr.lib.gsr_last_error()
    with pytest.raises(_gsr.GsrError):
        _gsr.set_option("bogus", 1)
    # profiling bookkeeping works without a device as long as nothing was recorded
    _gsr.profile_enable(False)
    assert all(v == (0.0, 0) for v in _gsr.profile_collect().values())


def test_missing_library_fails_loudly(tmp_path, monkeypatch):
    """There is no CPU fallback: importing the binding without the .so raises ImportError."""
    import importlib.util
    src = os.path.join(ROOT, "gaussian-splatting-reflection_amd", "_gsr.py")
    dst = tmp_path / "_gsr_copy.py"
    dst.write_text(open(src).read())
    spec = importlib.util.spec_from_file_location("_gsr_copy", str(dst))
    mod = importlib.util.module_from_spec(spec)
    with pytest.raises(ImportError, match="no CPU fallback|not found"):
        spec.loader.exec_module(mod)
