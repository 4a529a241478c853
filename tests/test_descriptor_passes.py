"""The descriptor passes of the hierarchical stage (AddOverSegmentation: sparse Lab histogram and flow
histograms of every region, video_segment_amd/csrc/region_segmentation.cpp) against their per-pixel form,
on the CPU.

The bytes of the hierarchical SegmentationDesc (tests/test_region_segmentation.py) only show a histogram
error that changes a merge; here the histograms themselves are compared: ColorHist::AddLabPixels (batched,
register sums over runs) with AddLabPixel pixel by pixel, FlowSamples + AccumulateFlow with FlowHist::Add
-- every bin bit for bit, the order of the sparse bins, weight sums, vector counts
(tests/host/descriptor_model.inc, compiled into region_segmentation.cpp with -DVSG_TEST_MODELS; g++, no HIP
call, no oracle)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "video_segment_amd", "csrc")


@pytest.fixture(scope="module")
def harness(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("desc") / "descriptor_harness")
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    subprocess.run(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-DVSG_TEST_MODELS", "-D__HIP_PLATFORM_AMD__",
                    "-I", os.path.join(rocm, "include"), "-I", CSRC,
                    os.path.join(ROOT, "tests", "host", "descriptor_harness.cpp"),
                    os.path.join(CSRC, "region_segmentation.cpp"), os.path.join(CSRC, "postprocess.cpp"),
                    os.path.join(CSRC, "boundary.cpp"), "-o", exe, "-pthread"], check=True, timeout=600)
    return exe


@pytest.mark.parametrize("seed", [1, 2])
def test_batched_passes_equal_per_pixel_passes(harness, seed):
    r = subprocess.run([harness, "400", str(seed)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "400 cases identical" in r.stdout
