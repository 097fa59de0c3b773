"""The C-ABI library loads without a GPU and exports every symbol include/vipsy_amd.h declares."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build():
    import __graft_entry__ as g
    g.build()


def test_library_exports_every_declared_symbol():
    _build()
    from vipsy_amd import _hip
    hdr = open(os.path.join(ROOT, "include", "vipsy_amd.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = sorted(set(re.findall(r"\b(vx_[a-z0-9_]+)\s*\(", hdr)))
    assert len(declared) >= 10
    handle = ctypes.CDLL(_hip.LIB_PATH)
    for name in declared:
        assert hasattr(handle, name), "library does not export %s" % name
    # the ctypes binding covers exactly the declared surface
    assert sorted(_hip.SIGNATURES) == declared
    lib = _hip.lib()
    assert lib.vx_abi_version() == _hip.ABI_VERSION
    assert b"gfx950" in lib.vx_build_info()


def test_struct_layout_matches_header():
    from vipsy_amd import _hip
    assert ctypes.sizeof(_hip.IrtCfg) == 72          # 4*int32 + 2*float + uint64 + 2*uint32 + the device step pointer + the rows ring
    assert _hip.IrtCfg.rows_ring.offset == 48 and _hip.IrtCfg.rows_ring_slots.offset == 64
    assert ctypes.sizeof(_hip.AdamSeg) == 24


def test_product_path_refuses_to_run_without_gpu():
    import torch
    from vipsy_amd import _hip
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(_hip.VxError):
        from vipsy_amd.engine import HipBackend
        HipBackend()
