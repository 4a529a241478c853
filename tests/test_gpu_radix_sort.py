"""The merge path's stable (key, value) sort -- the hand-written radix sort (csrc/radix_sort.hip), the
library's, and the choice the merge makes between them by size -- against numpy's stable sort.

Sizes on both sides of every switch of the implementation (one workgroup up to 4096 pairs, tile
histograms + a scanned histogram matrix above; the merge's window of sizes for the hand-written form), key widths
that give one to four passes, skewed and constant keys, bits above `end_bit` set (they must not take
part in the order but must travel with the key)."""
import ctypes as C

import numpy as np
import pytest

from test_gpu_parity import vsg  # noqa: F401  (fixture)

pytestmark = pytest.mark.gpu


def _sort(n, keys, vals, end_bit, impl=1):
    """impl 1: hand-written, 2: rocPRIM, 0: what the merge uses for n."""
    from video_segment_amd import _lib
    L = _lib.lib()
    ko = np.empty(max(n, 1), np.uint32)[:n]
    vo = np.empty(max(n, 1), np.uint32)[:n]
    rc = L.vsg_debug_sort_pairs_timed(keys.ctypes.data_as(C.c_void_p), vals.ctypes.data_as(C.c_void_p), n, end_bit,
                                      ko.ctypes.data_as(C.c_void_p), vo.ctypes.data_as(C.c_void_p), 0, impl, 1, None)
    assert rc == 0, L.vsg_last_error()
    return ko, vo


def _check(rng, n, end_bit, kind, impls=(1,)):
    if kind == "uniform":
        keys = rng.integers(0, 1 << end_bit, n, dtype=np.uint64).astype(np.uint32)
    elif kind == "few":          # long runs of equal keys: the order inside a run is the test
        keys = rng.integers(0, 5, n, dtype=np.uint64).astype(np.uint32) * np.uint32(max(1, (1 << end_bit) // 7))
    elif kind == "const":
        keys = np.full(n, (1 << end_bit) - 1, np.uint32)
    elif kind == "sorted_desc":
        keys = (np.arange(n, 0, -1, dtype=np.uint64) % (1 << end_bit)).astype(np.uint32)
    else:                         # bits above end_bit set at random
        keys = rng.integers(0, 1 << 32, n, dtype=np.uint64).astype(np.uint32)
    vals = rng.integers(0, 1 << 32, n, dtype=np.uint64).astype(np.uint32)
    mask = np.uint32((1 << end_bit) - 1) if end_bit < 32 else np.uint32(0xFFFFFFFF)
    order = np.argsort(keys & mask, kind="stable")
    for impl in impls:
        ko, vo = _sort(n, keys, vals, end_bit, impl)
        assert np.array_equal(ko, keys[order]), (n, end_bit, kind, impl)
        assert np.array_equal(vo, vals[order]), (n, end_bit, kind, impl)


@pytest.mark.parametrize("n", [1, 2, 63, 64, 65, 1000, 4095, 4096, 4097, 8192, 70001, 64 * 4096, 64 * 4096 + 1,
                               1 << 20, 3_000_017])
def test_sizes(vsg, n):
    rng = np.random.default_rng(n)
    for end_bit, kind in [(26, "uniform"), (21, "few"), (9, "high_bits"), (27, "sorted_desc")]:
        _check(rng, n, end_bit, kind)


@pytest.mark.parametrize("end_bit", [1, 5, 8, 9, 10, 17, 18, 19, 23, 27, 28, 32])
def test_key_widths(vsg, end_bit):
    rng = np.random.default_rng(end_bit)
    for n in (777, 50_000, 600_000):
        for kind in ("uniform", "const", "high_bits"):
            _check(rng, n, end_bit, kind)


def test_empty(vsg):
    k = np.zeros(1, np.uint32)
    _sort(0, k, k, 20)


def test_dispatch_and_library(vsg, monkeypatch):
    """The merge's own choice (impl 0) on both sides of its window, with the window moved and emptied, and the
    library alone."""
    rng = np.random.default_rng(3)
    for window in (None, "0:2000000000", "1:0", "5000:100000"):
        if window is None:
            monkeypatch.delenv("VSG_SORT_HAND", raising=False)
        else:
            monkeypatch.setenv("VSG_SORT_HAND", window)
        for n in (3000, 50_000, 450_000, 2_100_000):
            _check(rng, n, 26, "uniform", impls=(0, 2))
