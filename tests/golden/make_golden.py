"""Generates the golden vectors under tests/golden/ from the CPU oracle.

Provenance: the reference has no fixtures for this path and cannot be executed in this image, so
these vectors are produced by oracle/libvs_oracle.so, which is itself pinned bit-exactly to the
reference-derived values recorded in SURVEY.md App. B (tests/test_oracle_pins.py).  The vectors
freeze the complete serialized output (every SegmentationDesc byte) for inputs that the survey's
probe did not cover: noise, constant colour, single frame, the bench generator, L1 distance,
no pre-smoothing, many short chunks.

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


def main():
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
