"""Deterministic synthetic inputs shared by tests, bench.py and the golden-fixture scripts.

* ``probe_frame``   -- the survey probe driver's input: integer gradient +
  moving 16x12 checker, no noise.  Used for the oracle pins.
* ``bench_frame``   -- the BASELINE/SURVEY 8(d) bench input: gradient + moving checker scaled
  with the frame width + additive +-3 PCG noise.
* ``const_flow``    -- constant backward flow (-2, 0).
* ``var_flow``      -- spatially varying backward flow: a rotation + zoom field around the frame
  centre that changes with the frame, extra motion on a quarter of the checker cells, and two
  bands of vectors that point far outside the frame (``flow_torch`` is its torch form).
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


def noise_frame(W, H, t, amp=40):
    """The bench shape on an input with MANY SMALL regions: the gradient of bench_frame (no checker)
    plus independent noise of +-amp per pixel and channel (pcg32(4321 + t))."""
    x = np.arange(W, dtype=np.int64)[None, :]
    y = np.arange(H, dtype=np.int64)[:, None]
    img = np.empty((H, W, 3), np.int64)
    img[..., 0] = (x * 255) // W
    img[..., 1] = (y * 255) // H
    img[..., 2] = 128
    r = _pcg32_stream(4321 + t, W * H * 3).astype(np.int64).reshape(H, W, 3)
    img += (r % (2 * amp + 1)) - amp
    return np.clip(img, 0, 255).astype(np.uint8)


def blobs_frame(W, H, t, cell=48):
    """Value noise: a coarse grid of random colours (one per cell x cell block, pcg32(777)), moved
    2 px per frame like the checker so that the constant flow matches, plus the +-3 noise of
    bench_frame -- thousands of medium-sized regions per frame instead of a dozen giant ones."""
    x = (np.arange(W, dtype=np.int64)[None, :] + 2 * t) // cell
    y = np.arange(H, dtype=np.int64)[:, None] // cell
    gw = (W + 2 * 4096) // cell + 2
    key = (y * gw + x).astype(np.uint32)
    img = np.empty((H, W, 3), np.int64)
    for c in range(3):
        img[..., c] = _pcg_hash(key * np.uint32(3) + np.uint32(c) + _pcg_hash(np.array([777], np.uint32))[0]) % 224 + 16
    r = _pcg32_stream(1234 + t, W * H * 3).astype(np.int64).reshape(H, W, 3)
    img += (r % 7) - 3
    return np.clip(img, 0, 255).astype(np.uint8)


def soft_frame(W, H, t):
    """Input for the hierarchical stage: the bench generator with a LOW-contrast checker (R 110 /
    140 instead of 40 / 200).  Neighbouring regions then always share colour-histogram bins; on
    the bench input itself neighbouring checker cells have disjoint Lab histograms, their region
    distance is exactly 1.0 and the reference's RegionAgglomerationGraph aborts (a glog CHECK,
    region_segmentation_graph.cpp:165; see oracle/vs_oracle_region.inc)."""
    x = np.arange(W, dtype=np.int64)[None, :]
    y = np.arange(H, dtype=np.int64)[:, None]
    cw = max(1, (16 * W) // 64)
    ch = max(1, (12 * W) // 64)
    img = np.empty((H, W, 3), np.int64)
    img[..., 0] = 96 + (x * 64) // W
    img[..., 1] = 96 + (y * 64) // H
    chk = (((x + 2 * t) // cw) % 2) ^ ((y // ch) % 2)
    img[..., 2] = chk * 30 + 110
    r = _pcg32_stream(1234 + t, W * H * 3).astype(np.int64).reshape(H, W, 3)
    img += (r % 7) - 3
    return np.clip(img, 0, 255).astype(np.uint8)


def _pcg_hash_t(v):
    """_pcg_hash on a torch int64 tensor holding uint32 values."""
    m = 0xFFFFFFFF
    state = (v * 747796405 + 2891336453) & m
    word = (((state >> ((state >> 28) + 4)) ^ state) * 277803737) & m
    return ((word >> 22) ^ word) & m


def frame_torch(kind, W, H, t, device):
    """bench_frame / noise_frame / blobs_frame generated on `device` with torch (bit-identical to
    the numpy versions, tests/test_synth.py): the numpy generators need 0.15 s per 1080p frame."""
    import torch
    x = torch.arange(W, dtype=torch.int64, device=device)[None, :]
    y = torch.arange(H, dtype=torch.int64, device=device)[:, None]
    img = torch.empty((H, W, 3), dtype=torch.int64, device=device)

    def stream(seed):
        base = int(_pcg_hash(np.array([seed], np.uint32))[0])
        idx = (torch.arange(W * H * 3, dtype=torch.int64, device=device) + base) & 0xFFFFFFFF
        return _pcg_hash_t(idx).reshape(H, W, 3)

    if kind == "bench":
        cw = max(1, (16 * W) // 64)
        ch = max(1, (12 * W) // 64)
        img[..., 0] = ((x * 255) // W).expand(H, W)
        img[..., 1] = ((y * 255) // H).expand(H, W)
        chk = (((x + 2 * t) // cw) % 2) ^ ((y // ch) % 2)
        img[..., 2] = chk * 160 + 40
        img += (stream(1234 + t) % 7) - 3
    elif kind == "noise":
        amp = 40
        img[..., 0] = ((x * 255) // W).expand(H, W)
        img[..., 1] = ((y * 255) // H).expand(H, W)
        img[..., 2] = 128
        img += (stream(4321 + t) % (2 * amp + 1)) - amp
    elif kind == "soft":
        cw = max(1, (16 * W) // 64)
        ch = max(1, (12 * W) // 64)
        img[..., 0] = (96 + (x * 64) // W).expand(H, W)
        img[..., 1] = (96 + (y * 64) // H).expand(H, W)
        chk = (((x + 2 * t) // cw) % 2) ^ ((y // ch) % 2)
        img[..., 2] = chk * 30 + 110
        img += (stream(1234 + t) % 7) - 3
    elif kind == "blobs":
        cell = 48
        gx = (x + 2 * t) // cell
        gy = y // cell
        gw = (W + 2 * 4096) // cell + 2
        key = (gy * gw + gx) & 0xFFFFFFFF
        salt = int(_pcg_hash(np.array([777], np.uint32))[0])
        for c in range(3):
            img[..., c] = _pcg_hash_t((key * 3 + c + salt) & 0xFFFFFFFF) % 224 + 16
        img += (stream(1234 + t) % 7) - 3
    else:
        raise ValueError(kind)
    return img.clamp_(0, 255).to(torch.uint8)


FRAME_FNS = {"bench": bench_frame, "noise": noise_frame, "blobs": blobs_frame, "soft": soft_frame}


def const_flow(W, H, fx=-2.0, fy=0.0):
    f = np.empty((H, W, 2), np.float32)
    f[..., 0] = fx
    f[..., 1] = fy
    return f


def _var_flow_int(xp, x, y, W, H, t, hash_fn):
    """The flow of var_flow in units of 1/64 px as two int64 arrays (same integer operations in
    numpy and torch; every numerator is made non-negative before a division)."""
    cx, cy = W // 2, H // 2
    a = (t * 7) % 11 - 5          # zoom term of frame t, -5..5
    b = (t * 5) % 13 - 6          # rotation term, -6..6
    half = max(W // 2, 1)
    big = 1 << 20                 # offset that keeps the numerators positive (|num| < 6 * 1.2 * 192 * W)
    den = 5 * half
    # rotation + zoom: up to about +-6 px in the frame corners, on top of the (-2, 0) of const_flow
    fx = -128 + (a * (x - cx) * 192 - b * (y - cy) * 192 + big * den) // den - big
    fy = (b * (x - cx) * 192 + a * (y - cy) * 192 + big * den) // den - big
    # per-object motion: a quarter of the checker cells of bench_frame (as they lie in frame t) move
    # by up to +-2 px on their own
    cw = max(1, (16 * W) // 64)
    ch = max(1, (12 * W) // 64)
    cell = ((y // ch) * 4099 + (x + 2 * t) // cw) & 0xFFFFFFFF
    hsh = hash_fn(cell)
    moving = (hsh % 4) == 0
    fx = fx + xp.where(moving, ((hsh >> 2) % 5 - 2) * 64, 0)
    fy = fy + xp.where(moving, ((hsh >> 5) % 5 - 2) * 64, 0)
    # out-of-range bands (the reference clamps the displaced position, dense_segmentation_graph.h:1126-1135):
    # eight rows pointing 3 W to the right, eight columns pointing 3 H up
    row_band = (y >= H // 3) & (y < H // 3 + 8)
    col_band = (x >= (2 * W) // 3) & (x < (2 * W) // 3 + 8)
    fx = xp.where(row_band, 3 * W * 64, fx)
    fy = xp.where(col_band & ~row_band, -3 * H * 64, fy)
    return fx, fy


def var_flow(W, H, t):
    """Deterministic spatially varying backward flow of frame t (H x W x 2 f32, multiples of 1/64):
    smooth rotation / zoom field + per-object motion + a band of out-of-range vectors."""
    x = np.broadcast_to(np.arange(W, dtype=np.int64)[None, :], (H, W))
    y = np.broadcast_to(np.arange(H, dtype=np.int64)[:, None], (H, W))
    fx, fy = _var_flow_int(np, x, y, W, H, t, lambda v: _pcg_hash(v.astype(np.uint32)).astype(np.int64))
    f = np.empty((H, W, 2), np.float32)
    f[..., 0] = fx.astype(np.float32) * np.float32(1.0 / 64.0)
    f[..., 1] = fy.astype(np.float32) * np.float32(1.0 / 64.0)
    return f


def flow_torch(kind, W, H, t, device):
    """const_flow / var_flow generated on `device` (bit-identical to the numpy forms, tests/test_synth.py)."""
    import torch
    if kind == "const":
        f = torch.empty((H, W, 2), dtype=torch.float32, device=device)
        f[..., 0] = -2.0
        f[..., 1] = 0.0
        return f
    if kind != "var":
        raise ValueError(kind)
    x = torch.arange(W, dtype=torch.int64, device=device)[None, :].expand(H, W)
    y = torch.arange(H, dtype=torch.int64, device=device)[:, None].expand(H, W)
    fx, fy = _var_flow_int(torch, x, y, W, H, t, _pcg_hash_t)
    f = torch.empty((H, W, 2), dtype=torch.float32, device=device)
    f[..., 0] = fx.to(torch.float32) * (1.0 / 64.0)
    f[..., 1] = fy.to(torch.float32) * (1.0 / 64.0)
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
