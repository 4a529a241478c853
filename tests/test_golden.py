"""Golden vectors (tests/golden/*.json, produced by tests/golden/make_golden.py): the oracle must
keep reproducing them (CPU), and the product must reproduce them too -- the dense stream and its
vectorisation on the GPU, the hierarchical stage (host code of the product library) on the CPU and
behind the GPU dense unit."""
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


VGOLD = json.load(open(os.path.join(HERE, "golden", "vector_golden.json")))
HGOLD = json.load(open(os.path.join(HERE, "golden", "hierarchy_golden.json")))


@pytest.mark.parametrize("name", sorted(VGOLD))
def test_oracle_reproduces_vector_golden(name):
    g = VGOLD[name]
    opts = dict(chunk_size=g["chunk"], **g["options"])
    digests, lhash = make_golden.run_case(
        lambda: ol.OracleStream(g["W"], g["H"], ol.default_options(**opts), has_flow=g["flow"]),
        g["W"], g["H"], g["N"], g["kind"], g["flow"])
    assert lhash == g["label_fnv1a32"]
    assert digests == g["sha256_per_frame"]


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(VGOLD))
def test_hip_reproduces_vector_golden(name):
    """f2: Region2D.vectorization + vector_mesh bytes of the dense unit (compute_vectorization)."""
    import video_segment_amd as vsg
    g = VGOLD[name]
    opts = dict(chunk_size=g["chunk"], **g["options"])
    digests, lhash = make_golden.run_case(
        lambda: vsg.DenseSegmentation(g["W"], g["H"], vsg.default_options(**opts), has_flow=g["flow"]),
        g["W"], g["H"], g["N"], g["kind"], g["flow"])
    assert lhash == g["label_fnv1a32"]
    assert digests == g["sha256_per_frame"]


@pytest.mark.parametrize("name", sorted(HGOLD))
@pytest.mark.parametrize("impl", ["oracle", "product"])
def test_hierarchy_golden(name, impl):
    """f3: every hierarchy level of every frame, from the oracle's restatement and from the product's
    host implementation (vsg_regionseg_*), over the same oracle over-segmentation."""
    g = HGOLD[name]
    if impl == "oracle":
        make = lambda: ol.OracleRegionSegmentation(g["W"], g["H"], ol.region_options(**g["region_options"]))  # noqa: E731
    else:
        import video_segment_amd as vsg
        from video_segment_amd import _lib
        _lib.build()
        make = lambda: vsg.RegionSegmentation(g["W"], g["H"], vsg.default_region_options(**g["region_options"]))  # noqa: E731
    digests = make_golden.run_hierarchy_case(make, g["W"], g["H"], g["N"], g["chunk"], g["flow"])
    assert digests == g["sha256_per_frame"]
