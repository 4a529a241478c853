"""The C++ host mirror (VideoUnit tree: synthetic source -> DenseSegmentationUnit -> sink) run as the
reference's seg_tree_sample would be, checked against the reference-derived pins."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "video_segment_amd", "host")


def build():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "video_segment_amd", "csrc"), "-j8", "-s"])
    subprocess.check_call(["make", "-C", HOST, "-s"])


def test_host_driver_builds_and_fails_loudly_without_gpu():
    build()
    from video_segment_amd import _lib
    if _lib.lib().vsg_device_count() > 0:
        pytest.skip("a GPU is present")
    p = subprocess.run([os.path.join(HOST, "seg_tree_synth"), "--frames", "2"], capture_output=True,
                       text=True)
    assert p.returncode == 1
    assert "no usable HIP device" in p.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("args,frames,first,total,lhash", [
    (["--width", "64", "--height", "48", "--frames", "8", "--flow", "0"], 8, 264, 2222, "39aeeabb"),
    (["--width", "64", "--height", "48", "--frames", "45", "--flow", "1"], 45, 240, 12385, "5ef008e2"),
])
def test_host_unit_reproduces_reference_pins(args, frames, first, total, lhash):
    build()
    p = subprocess.run([os.path.join(HOST, "seg_tree_synth")] + args, capture_output=True, text=True,
                       timeout=300)
    assert p.returncode == 0, p.stderr
    m = re.search(r"frames=(\d+) first_frame_regions=(\d+) total_regions=(\d+) label_fnv1a32=(\w+)",
                  p.stdout)
    assert m, p.stdout
    assert (int(m.group(1)), int(m.group(2)), int(m.group(3)), m.group(4)) == (frames, first, total, lhash)
    assert "__STREAMING_SIZE__: %d" % frames in p.stderr
    assert "__SEGMENTATION_FINISHED__" in p.stderr
