"""Device time per call of the merge path's (key, value) sort, hand-written (csrc/radix_sort.hip) against
rocPRIM, over the sizes and key widths the merge uses.  python tools/sort_probe.py [reps]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from video_segment_amd import _lib

_lib.build()
L = _lib.lib()
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rng = np.random.default_rng(1)
print("%10s %4s %8s | %10s %10s" % ("n", "bits", "keys", "hand us", "rocPRIM us"))
for n in (256, 1024, 4096, 8192, 16384, 65536, 262144, 524288, 1 << 20, 4 << 20, 16 << 20, 60_000_000):
    for bits, kind in ((26, "uniform"), (26, "runs"), (22, "uniform")):
        if kind == "uniform":
            keys = rng.integers(0, 1 << bits, n, dtype=np.uint64).astype(np.uint32)
        else:   # component keys: runs of equal keys of very different lengths
            lens = np.minimum(rng.zipf(1.5, n), 50_000)
            keys = np.repeat(rng.integers(0, 1 << bits, n, dtype=np.uint64).astype(np.uint32), lens)[:n]
            keys = keys[rng.permutation(n)] if n < (1 << 22) else keys
        vals = np.arange(n, dtype=np.uint32)
        ko = np.empty(n, np.uint32)
        vo = np.empty(n, np.uint32)
        out = []
        for impl in (1, 2):
            us = C.c_double(0)
            rc = L.vsg_debug_sort_pairs_timed(keys.ctypes.data_as(C.c_void_p), vals.ctypes.data_as(C.c_void_p), n,
                                              bits, ko.ctypes.data_as(C.c_void_p), vo.ctypes.data_as(C.c_void_p), 0,
                                              impl, reps if n < (1 << 22) else 3, C.byref(us))
            assert rc == 0, L.vsg_last_error()
            out.append(us.value)
        print("%10d %4d %8s | %10.1f %10.1f" % (n, bits, kind, out[0], out[1]), flush=True)
