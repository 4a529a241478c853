"""Pins the CPU oracle against values produced by the reference's own code.

The reference has no tests or golden vectors for this path.  The only reference-derived
numbers available are the ones the survey session recorded while running the reference's
DenseSegmentation on a synthetic probe input (SURVEY.md Appendix B, table "Observed"):
number of Region2D in frame 0, number of hierarchy(0) regions in chunk 0, total Region2D over
all output frames, FNV-1a-32 of all output label planes, and the first Region2D's moments.
The oracle must reproduce every one of them bit-exactly.
"""
import numpy as np
import pytest

import oracle_lib as ol
import synth

# (W, H, N, flow) -> (frame-0 Region2D, hierarchy(0) regions, sum Region2D, label hash)
PINS = {
    (64, 48, 8, False): (264, 1105, 2222, 0x39AEEABB),
    (64, 48, 8, True): (240, 1319, 2290, 0xE71BAB4E),
    (64, 48, 45, True): (240, 3520, 12385, 0x5EF008E2),
    (320, 240, 22, True): (224, 380, 5232, 0xE29D154D),
    (640, 480, 22, False): (1988, 2558, 44296, 0x8D5C857C),
    (1920, 1080, 22, True): (90, 90, 1991, 0xDF411091),
}


def run_probe(W, H, N, flow):
    s = ol.OracleStream(W, H, ol.default_options(), has_flow=flow)
    fl = synth.const_flow(W, H) if flow else None
    planes, counts, hier0, first = [], [], None, None
    for k in range(N):
        n = s.process_frame(synth.probe_frame(W, H, k), fl if (flow and k > 0) else None,
                            flush=(k == N - 1))
        for i in range(n):
            planes.append(s.result_id_image(i))
            counts.append(s.result_num_regions(i))
            if hier0 is None:
                hier0 = s.result_hierarchy_regions(i)
                first = s.result_first_region(i)
    s.close()
    return planes, counts, hier0, first


def _check(key):
    planes, counts, hier0, first = run_probe(*key)
    f0, h0, total, lhash = PINS[key]
    assert len(planes) == key[2]
    assert counts[0] == f0
    assert hier0 == h0
    assert sum(counts) == total
    assert synth.fnv1a32_fast(planes) == lhash
    return first


@pytest.mark.parametrize("key", [(64, 48, 8, False), (64, 48, 8, True), (64, 48, 45, True),
                                 (320, 240, 22, True)])
def test_oracle_matches_reference_probe(key):
    first = _check(key)
    if key == (64, 48, 8, False):
        rid, m = first
        # "id 0, size 8, mean (1.5, 0.5), moment_xx 3.5".
        assert rid == 0 and m[0] == 8.0 and m[1] == 1.5 and m[2] == 0.5 and m[3] == 3.5


def test_oracle_matches_reference_probe_vga():
    _check((640, 480, 22, False))


@pytest.mark.slow
def test_oracle_matches_reference_probe_1080p():
    rid, m = _check((1920, 1080, 22, True))
    # "id 0, size 48086, mean (666.239624, 35.610695), moment_xx 592062.9375".
    assert rid == 0 and m[0] == 48086.0
    assert m[1] == np.float32(666.239624) and m[2] == np.float32(35.610695)
    assert m[3] == np.float32(592062.9375)


def test_oracle_matches_reference_config2_probe():
    """SURVEY App. B, 'Config-2-shaped probe through seam (3)': DenseSegmentationGraph(640,480,32),
    32 x {convertTo, BilateralFilter, AddNodesAndSpatialEdges}, SegmentFullGraph(983, false),
    ObtainResults(nullptr, false, true, true), DetermineNeighborIds -> 288 regions, 1184 directed
    neighbour links."""
    W, H, F = 640, 480, 32
    g = ol.OracleGraph(W, H, F)
    for k in range(F):
        g.add_frame(ol.preprocess(synth.probe_frame(W, H, k)))
    g.segment(983, False)
    g.obtain_results(None, True, True)
    assert (g.num_regions(), g.num_neighbor_links()) == (288, 1184)
