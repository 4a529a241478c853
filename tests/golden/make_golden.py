"""Generates the golden vectors under tests/golden/ from the CPU oracle.

Provenance: the reference has no fixtures for this path and cannot be executed in this image, so
these vectors are produced by oracle/libvs_oracle.so, which is itself pinned bit-exactly to the
reference-derived values recorded in SURVEY.md App. B (tests/test_oracle_pins.py).  The vectors
freeze the complete serialized output (every SegmentationDesc byte) for inputs that the survey's
probe did not cover: noise, constant colour, single frame, the bench generator, L1 distance,
no pre-smoothing, Gaussian pre-smoothing, many short chunks; and, in their own files, for the two stages on the caller side
of the path -- the boundary vectorisation of the dense unit (vector_golden.json: every byte of
Region2D.vectorization and vector_mesh) and the hierarchical RegionSegmentation behind it
(hierarchy_golden.json: every hierarchy level, two and three chunk sets with overlap and
constraints).  Both are "parity unpinned" like the oracle code they come from:
they freeze today's bytes against regressions, they do not anchor them to the reference.

    python tests/golden/make_golden.py
"""
import hashlib
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))

import numpy as np  # noqa: E402
import oracle_lib as ol  # noqa: E402
import synth  # noqa: E402

CASES = [
    # name, W, H, N, kind, flow, chunk, options
    ("probe_64x48x8", 64, 48, 8, "probe", False, 20, {}),
    ("probe_flow_64x48x45", 64, 48, 45, "probe", True, 20, {}),
    ("bench_96x64x30_c10", 96, 64, 30, "bench", True, 10, {}),
    ("noise_48x40x20_c8", 48, 40, 20, "noise", True, 8, {}),
    ("const_50x36x12", 50, 36, 12, "const", False, 20, {}),
    ("single_frame_64x48", 64, 48, 1, "probe", False, 20, {}),
    ("bench_l1_64x48x12", 64, 48, 12, "bench", True, 8, {"color_distance": 0}),
    ("bench_nosmooth_64x48x12", 64, 48, 12, "bench", True, 8, {"presmoothing": 0}),
    ("bench_gaussian_64x48x20", 64, 48, 20, "bench", True, 8, {"presmoothing": 1}),
]


def frame_of(kind, W, H, k, rng):
    if kind == "probe":
        return synth.probe_frame(W, H, k)
    if kind == "bench":
        return synth.bench_frame(W, H, k)
    if kind == "noise":
        return rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    if kind == "const":
        return np.full((H, W, 3), 77, np.uint8)
    raise ValueError(kind)


def run_case(make_stream, W, H, N, kind, flow):
    rng = np.random.default_rng(42)
    s = make_stream()
    fl = synth.const_flow(W, H) if flow else None
    digests, planes = [], []
    for k in range(N):
        n = s.process_frame(frame_of(kind, W, H, k, rng), fl if (flow and k > 0) else None,
                            flush=(k == N - 1))
        for i in range(n):
            digests.append(hashlib.sha256(s.result_bytes(i)).hexdigest())
            planes.append(s.result_id_image(i))
    s.close()
    return digests, "%08x" % synth.fnv1a32_fast(planes)


# ---- f2: dense stream with compute_vectorization ----------------------------------------------------
VECTOR_CASES = [
    ("vector_bench_96x64x20_c8", 96, 64, 20, "bench", True, 8, {"compute_vectorization": 1}),
    ("vector_probe_64x48x12_c8", 64, 48, 12, "probe", False, 8, {"compute_vectorization": 1}),
    ("vector_noise_nosmooth_48x40x9_c8", 48, 40, 9, "noise", True, 8,
     {"compute_vectorization": 1, "presmoothing": 0}),
]

# ---- f3: hierarchical RegionSegmentation over the oracle's over-segmentation ---------------------------
HIERARCHY_CASES = [
    # name, W, H, N, chunk, flow, region options
    ("hier_soft_96x64x60_c8_sets3", 96, 64, 60, 8, True,
     dict(chunk_set_size=3, chunk_set_overlap=1, constraint_chunks=1, min_region_num=3)),
    ("hier_soft_80x60x24_c8_noflow", 80, 60, 24, 8, False, dict(use_flow=0, min_region_num=4)),
    ("hier_soft_96x64x30_c10_cut", 96, 64, 30, 10, True,
     dict(chunk_set_size=2, chunk_set_overlap=1, max_region_num=20, min_region_num=3)),
    ("hier_soft_96x64x40_c8_features", 96, 64, 40, 8, True,
     dict(chunk_set_size=3, chunk_set_overlap=1, min_region_num=3, save_descriptors=1)),
]


def run_hierarchy_case(make_region_seg, W, H, N, chunk, flow):
    """Dense over-segmentation by the oracle stream (soft generator: the bench checker makes the
    reference abort), hierarchy by `make_region_seg()`; sha256 of every result."""
    fl = synth.const_flow(W, H) if flow else None
    o = ol.OracleStream(W, H, ol.default_options(chunk_size=chunk), has_flow=flow)
    frames = [synth.soft_frame(W, H, k) for k in range(N)]
    seg = []
    for k in range(N):
        n = o.process_frame(frames[k], fl if (flow and k > 0) else None, flush=(k == N - 1))
        seg += [o.result_bytes(i) for i in range(n)]
    o.close()
    r = make_region_seg()
    digests = []
    for k in range(N):
        n = r.process_frame(seg[k], frames[k], fl if (flow and k > 0) else None, flush=(k == N - 1))
        assert n >= 0
        digests += [hashlib.sha256(r.result_bytes(i)).hexdigest() for i in range(n)]
    r.close()
    return digests


def main():
    out = {}
    for name, W, H, N, kind, flow, chunk, extra in VECTOR_CASES:
        opts = dict(chunk_size=chunk, **extra)
        digests, lhash = run_case(
            lambda: ol.OracleStream(W, H, ol.default_options(**opts), has_flow=flow), W, H, N, kind, flow)
        out[name] = {"W": W, "H": H, "N": N, "kind": kind, "flow": flow, "chunk": chunk,
                     "options": extra, "label_fnv1a32": lhash, "sha256_per_frame": digests}
        print(name, lhash, len(digests))
    with open(os.path.join(HERE, "vector_golden.json"), "w") as f:
        json.dump(out, f, indent=1)
    out = {}
    for name, W, H, N, chunk, flow, ropts in HIERARCHY_CASES:
        digests = run_hierarchy_case(lambda: ol.OracleRegionSegmentation(W, H, ol.region_options(**ropts)),
                                     W, H, N, chunk, flow)
        out[name] = {"W": W, "H": H, "N": N, "chunk": chunk, "flow": flow, "region_options": ropts,
                     "sha256_per_frame": digests}
        print(name, len(digests))
    with open(os.path.join(HERE, "hierarchy_golden.json"), "w") as f:
        json.dump(out, f, indent=1)
    out = {}
    for name, W, H, N, kind, flow, chunk, extra in CASES:
        opts = dict(chunk_size=chunk, **extra)
        digests, lhash = run_case(
            lambda: ol.OracleStream(W, H, ol.default_options(**opts), has_flow=flow), W, H, N, kind, flow)
        out[name] = {"W": W, "H": H, "N": N, "kind": kind, "flow": flow, "chunk": chunk,
                     "options": extra, "label_fnv1a32": lhash, "sha256_per_frame": digests}
        print(name, lhash, len(digests))
    with open(os.path.join(HERE, "stream_golden.json"), "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
