"""Golden vectors (tests/golden/stream_golden.json, produced by tests/golden/make_golden.py):
the oracle must keep reproducing them (CPU), and the HIP path must reproduce them too (GPU)."""
import json
import os
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import make_golden  # noqa: E402
import oracle_lib as ol  # noqa: E402

GOLD = json.load(open(os.path.join(HERE, "golden", "stream_golden.json")))


@pytest.mark.parametrize("name", sorted(GOLD))
def test_oracle_reproduces_golden(name):
    g = GOLD[name]
    opts = dict(chunk_size=g["chunk"], **g["options"])
    digests, lhash = make_golden.run_case(
        lambda: ol.OracleStream(g["W"], g["H"], ol.default_options(**opts), has_flow=g["flow"]),
        g["W"], g["H"], g["N"], g["kind"], g["flow"])
    assert lhash == g["label_fnv1a32"]
    assert digests == g["sha256_per_frame"]


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(GOLD))
def test_hip_reproduces_golden(name):
    import video_segment_amd as vsg
    g = GOLD[name]
    opts = dict(chunk_size=g["chunk"], **g["options"])
    digests, lhash = make_golden.run_case(
        lambda: vsg.DenseSegmentation(g["W"], g["H"], vsg.default_options(**opts), has_flow=g["flow"]),
        g["W"], g["H"], g["N"], g["kind"], g["flow"])
    assert lhash == g["label_fnv1a32"]
    assert digests == g["sha256_per_frame"]
