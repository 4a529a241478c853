"""CPU-side checks of the C-ABI library: it loads, exports every symbol include/vsg.h declares,
and refuses loudly to run without a GPU (no fallback path)."""
import ctypes as C
import os
import re

import pytest

from video_segment_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def library():
    _lib.build()
    return _lib.lib()


def test_header_symbols_exported(library):
    header = open(os.path.join(ROOT, "include", "vsg.h")).read()
    declared = set(re.findall(r"\b(vsg_[a-z0-9_]+)\s*\(", header))
    assert declared, "no declarations parsed"
    assert declared == set(_lib.EXPORTED_SYMBOLS)
    for name in declared:
        assert hasattr(library, name), name


def test_default_options(library):
    o = _lib.VsgOptions()
    library.vsg_default_options(C.byref(o))
    assert (o.presmoothing, o.chunk_size, o.color_distance) == (2, 20, 1)
    assert abs(o.frac_min_region_size - 0.01) < 1e-7 and abs(o.chunk_overlap_ratio - 0.2) < 1e-7
    assert library.vsg_version() >= 100


def test_no_cpu_fallback(library):
    """Without a HIP device every create call must fail with VSG_ERR_DEVICE."""
    if library.vsg_device_count() > 0:
        pytest.skip("a GPU is present")
    o = _lib.VsgOptions()
    library.vsg_default_options(C.byref(o))
    h = C.c_void_p()
    assert library.vsg_stream_create(C.byref(o), 64, 48, C.byref(h)) == -2
    assert b"no usable HIP device" in library.vsg_last_error()
    assert library.vsg_graph_create(64, 48, 4, 0, -1, C.byref(h)) == -2
