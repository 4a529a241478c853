"""Deterministic synthetic inputs shared by tests, bench.py and the golden-fixture scripts.

* ``probe_frame``   -- the survey probe driver's input: integer gradient +
  moving 16x12 checker, no noise.  Used for the oracle pins.
* ``bench_frame``   -- the BASELINE/SURVEY 8(d) bench input: gradient + moving checker scaled
  with the frame width + additive +-3 PCG noise.
* ``const_flow``    -- constant backward flow (-2, 0).
"""
import numpy as np


def probe_frame(W, H, k):
    """B=(x*255)/W, G=(y*255)/H, R=((((x+2k)/16)%2) ^ ((y/12)%2)) ? 200 : 40  (integer ops)."""
    x = np.arange(W, dtype=np.int64)[None, :]
    y = np.arange(H, dtype=np.int64)[:, None]
    img = np.empty((H, W, 3), np.uint8)
    img[..., 0] = np.broadcast_to((x * 255) // W, (H, W))
    img[..., 1] = np.broadcast_to((y * 255) // H, (H, W))
    chk = (((x + 2 * k) // 16) % 2) ^ ((y // 12) % 2)
    img[..., 2] = np.where(chk != 0, 200, 40)
    return img


def _pcg_hash(v):
    """Counter-based PCG hash (pcg_hash, Jarzynski & Olano 2020), vectorised over uint32."""
    with np.errstate(over="ignore"):
        v = v.astype(np.uint32)
        state = v * np.uint32(747796405) + np.uint32(2891336453)
        word = ((state >> ((state >> np.uint32(28)) + np.uint32(4))) ^ state) * np.uint32(277803737)
        return (word >> np.uint32(22)) ^ word


def _pcg32_stream(seed, n):
    """n pseudo-random uint32: pcg_hash(pcg_hash(seed) + i)."""
    base = _pcg_hash(np.array([seed], np.uint32))[0]
    with np.errstate(over="ignore"):
        return _pcg_hash(np.arange(n, dtype=np.uint32) + base)


def bench_frame(W, H, t, noise=True):
    """bgr[t,y,x] = ((x*255)/W, (y*255)/H, checker(x+2t, y)*160+40) + pcg32(1234+t) in [-3,3].

    Checker cell is 16x12 px at 64x48 and scales with W/64."""
    x = np.arange(W, dtype=np.int64)[None, :]
    y = np.arange(H, dtype=np.int64)[:, None]
    cw = max(1, (16 * W) // 64)
    ch = max(1, (12 * W) // 64)
    img = np.empty((H, W, 3), np.int64)
    img[..., 0] = (x * 255) // W
    img[..., 1] = (y * 255) // H
    chk = (((x + 2 * t) // cw) % 2) ^ ((y // ch) % 2)
    img[..., 2] = chk * 160 + 40
    if noise:
        r = _pcg32_stream(1234 + t, W * H * 3).astype(np.int64).reshape(H, W, 3)
        img += (r % 7) - 3
    return np.clip(img, 0, 255).astype(np.uint8)


def const_flow(W, H, fx=-2.0, fy=0.0):
    f = np.empty((H, W, 2), np.float32)
    f[..., 0] = fx
    f[..., 1] = fy
    return f


def fnv1a32(arrays):
    """FNV-1a-32 over the little-endian bytes of the given int32 arrays (seed 2166136261)."""
    h = 2166136261
    for a in arrays:
        b = np.ascontiguousarray(a, dtype="<i4").tobytes()
        for byte in b:
            h = ((h ^ byte) * 16777619) & 0xFFFFFFFF
    return h


def fnv1a32_fast(arrays):
    """Same as fnv1a32 via a small C helper when available (hashing megabytes in Python is slow)."""
    try:
        import ctypes as C
        import os
        import subprocess
        import tempfile
        d = os.path.join(tempfile.gettempdir(), "vsg_fnv")
        so = os.path.join(d, "fnv.so")
        if not os.path.exists(so):
            os.makedirs(d, exist_ok=True)
            src = os.path.join(d, "fnv.c")
            with open(src, "w") as f:
                f.write("#include <stdint.h>\n#include <stddef.h>\n"
                        "uint32_t fnv(uint32_t h,const uint8_t*p,size_t n){"
                        "for(size_t i=0;i<n;++i){h^=p[i];h*=16777619u;}return h;}\n")
            subprocess.check_call(["gcc", "-O2", "-shared", "-fPIC", "-o", so, src])
        L = C.CDLL(so)
        L.fnv.restype = C.c_uint32
        L.fnv.argtypes = [C.c_uint32, C.c_void_p, C.c_size_t]
        h = 2166136261
        for a in arrays:
            b = np.ascontiguousarray(a, dtype="<i4")
            h = L.fnv(h, b.ctypes.data, b.nbytes)
        return h
    except Exception:
        return fnv1a32(arrays)
