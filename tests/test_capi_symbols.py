"""CPU-side checks of the drop-in boundary: the C-ABI library builds, loads and
exports every symbol include/lrp_hip.h declares (no compute calls without a GPU)."""
import ctypes
import os
import re

from conftest import ROOT


def _declared():
    src = open(os.path.join(ROOT, "include", "lrp_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(lrp_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from lrp_imagecaptioning_amd import _capi
    from lrp_imagecaptioning_amd.build import build_library
    build_library()
    lib = _capi.load()
    names = _declared()
    assert len(names) >= 15
    for n in names:
        assert hasattr(lib, n), n
        assert n in _capi.SYMBOLS, "binding missing for " + n
    assert sorted(_capi.SYMBOLS) == names
    assert lib.lrp_abi_version() == _capi.LRP_ABI_VERSION


def test_config_struct_layout_matches_header():
    from lrp_imagecaptioning_amd import _capi
    # 6 scalars + 3*32 ints + 32*32 chars + 5 + 3 + 2 ints + (ABI v2) 3 ints + 2*8 ints
    assert ctypes.sizeof(_capi.LrpConfig) == 4 * (6 + 3 * 32 + 5 + 3 + 2 + 3 + 16) + 32 * 32


def test_error_path_without_gpu():
    """Argument validation happens before any HIP call and reports through lrp_last_error."""
    from lrp_imagecaptioning_amd import _capi
    lib = _capi.load()
    h = ctypes.c_void_p()
    rc = lib.lrp_create(None, ctypes.byref(h))
    assert rc == _capi.LRP_ERR_INVALID
    assert b"null" in lib.lrp_last_error()
    cfg = _capi.LrpConfig()
    cfg.abi_version = 999
    assert lib.lrp_create(ctypes.byref(cfg), ctypes.byref(h)) == _capi.LRP_ERR_INVALID
    assert lib.lrp_destroy(None) == _capi.LRP_OK


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from lrp_imagecaptioning_amd import _capi
    monkeypatch.setattr(_capi, "_lib", None)
    monkeypatch.setattr(_capi, "LIB_PATH", str(tmp_path / "nope.so"))
    try:
        _capi.load()
    except _capi.LrpLibraryMissing as e:
        assert "no CPU fallback" in str(e)
    else:
        raise AssertionError("load() must raise when the HIP library is missing")


def test_no_exception_crosses_the_abi():
    """include/lrp_hip.h: 'return value: 0 = LRP_OK, negative = error' — a C++ exception must never unwind through an
    extern "C" entry into ctypes.  lrp_op_conv packs its weights into a std::vector sized from the caller's channel
    counts BEFORE any HIP call; 2^27 x 2^27 channels make that allocation (~650 PB) throw std::bad_alloc, which has to
    come back as LRP_ERR_NOMEM with a message (every entry point runs inside the same guard, csrc/engine.hip)."""
    import numpy as np
    from lrp_imagecaptioning_amd import _capi
    lib = _capi.load()
    dummy = np.zeros(16, dtype=np.float32)
    p = dummy.ctypes.data_as(ctypes.c_void_p)
    big = 1 << 27
    rc = lib.lrp_op_conv(p, p, p, None, p, 1, 1, 1, big, big, 9, 1, None)
    assert rc == _capi.LRP_ERR_NOMEM, rc
    assert b"memory" in lib.lrp_last_error()
    try:
        _capi.check(rc)
    except MemoryError:
        pass
    else:
        raise AssertionError("LRP_ERR_NOMEM must map to MemoryError")


def test_every_switch_of_the_library_is_listed_here():
    """csrc/common.h names the switches the library reads; a new one must come with a case above."""
    import os
    import re
    from conftest import ROOT
    from test_gpu_switches import CASES
    src = open(os.path.join(ROOT, "lrp-imagecaptioning_amd", "csrc", "common.h")).read()
    in_lib = set(re.findall(r'rd\("(LRP_[A-Z0-9_]+)"', src))
    assert in_lib == {c[0] for c in CASES}, in_lib ^ {c[0] for c in CASES}
    # nothing else in csrc reads the environment
    for f in os.listdir(os.path.join(ROOT, "lrp-imagecaptioning_amd", "csrc")):
        if f != "common.h":
            assert "getenv" not in open(os.path.join(ROOT, "lrp-imagecaptioning_amd", "csrc", f)).read(), f
