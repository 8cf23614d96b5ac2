"""The C-ABI library loads without a GPU and exports every symbol include/mhimx.h declares."""
import os
import re

from mhim_mil_amd import _lib as L

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "mhimx.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(mhimx_[a-z0-9_]+)\s*\(", src)))


def test_library_loads_and_exports_every_declared_symbol():
    lib = L.lib()
    names = _declared()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), f"libmhimx.so lacks {n}"
        assert n in L.SYMBOLS, f"ctypes binding lacks {n}"
    assert set(L.SYMBOLS) == set(names)
    assert lib.mhimx_version() == L.ABI_VERSION
    hdr = open(os.path.join(ROOT, "include", "mhimx.h")).read()
    assert int(re.search(r"#define MHIMX_VERSION (\d+)", hdr).group(1)) == L.ABI_VERSION


def test_argument_errors_are_reported_not_thrown():
    lib = L.lib()
    # no GPU needed: argument validation happens before any launch
    assert lib.mhimx_gemm_nt(None, None) < 0
    assert b"null" in lib.mhimx_last_error()
    assert lib.mhimx_abmil_pool_ws_bytes(1000, 512, 128, 0) > 0
    assert lib.mhimx_merge_ws_bytes(970, 512, 5, 8, 64) > 0
    assert lib.mhimx_select_ws_bytes(10000) >= 10000


def test_comm_handle_argument_errors():
    """mhimx_comm_*: argument validation needs neither a GPU nor RCCL; destroying a null handle is a no-op."""
    import ctypes as C
    lib = L.lib()
    assert lib.mhimx_comm_destroy(None) == 0
    h = C.c_void_p()
    assert lib.mhimx_comm_init(C.byref(h), None, 0, 1) < 0
    assert lib.mhimx_comm_init(C.byref(h), C.create_string_buffer(128), 3, 2) < 0
    assert lib.mhimx_comm_allreduce(None, None, None, 4, 0) < 0
    assert lib.mhimx_comm_unique_id(None) < 0


def test_binding_refuses_a_library_of_another_abi_version(monkeypatch):
    from mhim_mil_amd import _lib as L
    assert L.lib().mhimx_version() == L.ABI_VERSION
    monkeypatch.setattr(L, "_lib", None)
    monkeypatch.setattr(L, "ABI_VERSION", L.ABI_VERSION + 1)
    import pytest
    with pytest.raises(RuntimeError, match="ABI version"):
        L.lib()


def test_graft_entry_build_passes():
    """The driver's build check: compiles (or finds) the library, loads it, checks the ABI version and imports the package."""
    import __graft_entry__ as g
    g.build()
