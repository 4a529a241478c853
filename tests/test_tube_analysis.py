"""The host tube analysis (EnforceSpatialConnectedness, video_segment_amd/csrc/postprocess.cpp:
TubeSplitter) against its plain restatement, on the CPU.

Finish keeps the live tubes in a linked list, finds a tube's closest tube through per-frame lists and
evaluates a distance only where a bound says it can win; tests/host/tube_plain_model.inc is the same
analysis with a vector, erase and every distance evaluated, in the order of
segmentation/dense_segmentation_graph.h:666-861.  Both are compiled from postprocess.cpp with g++ (no
HIP, no oracle) and have to return the same tubes, areas and kept tube on random regions.  The same
run checks SplitComponentsN4 (a sweep over the previous row) against the components found by testing
every pair of intervals."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def harness(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("tube") / "tube_harness")
    subprocess.run(["g++", "-O2", "-std=c++17", "-DVSG_TEST_MODELS", "-I", os.path.join(ROOT, "video_segment_amd", "csrc"),
                    os.path.join(ROOT, "tests", "host", "tube_harness.cpp"),
                    os.path.join(ROOT, "video_segment_amd", "csrc", "postprocess.cpp"), "-o", exe, "-pthread"],
                   check=True, timeout=300)
    return exe


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_finish_equals_plain_restatement(harness, seed):
    r = subprocess.run([harness, "random", "700", str(seed)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "700 cases identical" in r.stdout
    # the cases have to reach both joins: regions that stay split and regions whose tubes are joined
    words = r.stdout.replace("(", " ").split()
    assert int(words[words.index("regions") - 1]) > 50 and int(words[words.index("joins)") - 1]) > 10000
