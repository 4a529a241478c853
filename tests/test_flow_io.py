"""`.flow` files: Python reader/writer round trip, the oracle stream fed
from a file, and the C++ DenseFlowReaderUnit of the host mirror feeding the HIP path (GPU)."""
import os
import re
import struct
import subprocess
import tempfile

import numpy as np
import pytest

import oracle_lib as ol
import synth
from video_segment_amd import flow_io

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "video_segment_amd", "host")


def field(W, H, k):
    rng = np.random.default_rng(100 + k)
    return (rng.standard_normal((H, W, 2)) * 2.5).astype(np.float32)


@pytest.mark.parametrize("flow_type", [flow_io.FLOW_FORWARD, flow_io.FLOW_BACKWARD, flow_io.FLOW_BOTH])
def test_flow_file_round_trip(flow_type):
    W, H, N = 20, 12, 6
    with tempfile.TemporaryDirectory() as d:
        p = os.path.join(d, "v.flow")
        w = flow_io.DenseFlowWriter(p)
        w.open_and_write_header(W, H, flow_type)
        for k in range(1, N):
            if flow_type == flow_io.FLOW_BOTH:
                w.add_flow_frame(-field(W, H, k))          # forward first
            w.add_flow_frame(field(W, H, k))
        w.close()
        per = 2 if flow_type == flow_io.FLOW_BOTH else 1
        assert os.path.getsize(p) == 12 + per * (N - 1) * W * H * 8
        assert struct.unpack("<iii", open(p, "rb").read(12)) == (W, H, flow_type)
        r = flow_io.DenseFlowReader(p)
        r.open_and_read_header()
        assert (r.width, r.height, r.flow_type) == (W, H, flow_type)
        backward = flow_type != flow_io.FLOW_FORWARD
        got = list(r.fields(backward=backward))
        r.close()
        assert got[0] is None and len(got) == N
        for k in range(1, N):
            assert np.array_equal(got[k], field(W, H, k))
        if flow_type != flow_io.FLOW_BOTH:
            r = flow_io.DenseFlowReader(p)
            with pytest.raises(ValueError):
                list(r.fields(backward=not backward))
            r.close()


def test_truncated_and_malformed_files():
    with tempfile.TemporaryDirectory() as d:
        p = os.path.join(d, "t.flow")
        open(p, "wb").write(struct.pack("<iii", 8, 4, 1) + b"\0" * 100)
        r = flow_io.DenseFlowReader(p)
        with pytest.raises(ValueError):
            list(r.fields())
        r.close()
        open(p, "wb").write(struct.pack("<iii", 8, -4, 1))
        with pytest.raises(ValueError):
            flow_io.DenseFlowReader(p).open_and_read_header()
        open(p, "wb").write(b"\1\0\0")
        with pytest.raises(ValueError):
            flow_io.DenseFlowReader(p).open_and_read_header()


def test_oracle_stream_from_flow_file_matches_in_memory_flow():
    """45 probe frames with the constant flow read back from a file reproduce the App. B pin."""
    W, H, N = 64, 48, 45
    with tempfile.TemporaryDirectory() as d:
        p = os.path.join(d, "probe.flow")
        w = flow_io.DenseFlowWriter(p)
        w.open_and_write_header(W, H, flow_io.FLOW_BACKWARD)
        for k in range(1, N):
            w.add_flow_frame(synth.const_flow(W, H))
        w.close()
        r = flow_io.DenseFlowReader(p)
        s = ol.OracleStream(W, H, ol.default_options(chunk_size=20), has_flow=True)
        ids = []
        for k, fl in zip(range(N), r.fields()):
            n = s.process_frame(synth.probe_frame(W, H, k), fl, flush=(k == N - 1))
            ids += [s.result_id_image(i) for i in range(n)]
        s.close()
        r.close()
        assert len(ids) == N
        assert "%08x" % synth.fnv1a32_fast(np.stack(ids)) == "5ef008e2"


@pytest.mark.gpu
def test_host_flow_reader_unit_feeds_the_hip_path():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "video_segment_amd", "csrc"), "-j8", "-s"])
    subprocess.check_call(["make", "-C", HOST, "-s"])
    W, H, N = 64, 48, 45
    exe = os.path.join(HOST, "seg_tree_synth")
    base = ["--width", str(W), "--height", str(H), "--frames", str(N), "--flow", "1"]
    with tempfile.TemporaryDirectory() as d:
        p = os.path.join(d, "probe.flow")
        a = subprocess.run([exe] + base + ["--flow_output_file", p], capture_output=True, text=True, timeout=300)
        assert a.returncode == 0, a.stderr
        assert os.path.getsize(p) == 12 + (N - 1) * W * H * 8
        r = flow_io.DenseFlowReader(p)
        got = list(r.fields())
        r.close()
        assert len(got) == N and all(np.array_equal(g, synth.const_flow(W, H)) for g in got[1:])
        b = subprocess.run([exe] + base + ["--flow_file", p], capture_output=True, text=True, timeout=300)
        assert b.returncode == 0, b.stderr
        ha = re.search(r"label_fnv1a32=(\w+)", a.stdout).group(1)
        hb = re.search(r"label_fnv1a32=(\w+)", b.stdout).group(1)
        assert ha == hb == "5ef008e2"
        # A flow file of the wrong size is refused at OpenStreams like the reference does.
        c = subprocess.run([exe, "--width", "32", "--height", "48", "--frames", "4", "--flow", "1",
                            "--flow_file", p], capture_output=True, text=True, timeout=300)
        assert c.returncode == 1 and "different dimension" in c.stderr


@pytest.mark.gpu
def test_raw_video_file_with_flow_file_next_to_it():
    """seg_tree_sample's wiring on files: <video> + <video base>.flow (seg_tree.cpp:120-169) through
    RawVideoReaderUnit -> DenseFlowReaderUnit -> DenseSegmentationUnit reproduces the App. B pin."""
    from video_segment_amd.raw_video import read_raw_video, write_raw_video
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "video_segment_amd", "csrc"), "-j8", "-s"])
    subprocess.check_call(["make", "-C", HOST, "-s"])
    W, H, N = 64, 48, 45
    exe = os.path.join(HOST, "seg_tree_synth")
    with tempfile.TemporaryDirectory() as d:
        video = os.path.join(d, "clip.rawv")
        write_raw_video(video, [synth.probe_frame(W, H, k) for k in range(N)])
        back, fps = read_raw_video(video)
        assert back.shape == (N, H, W, 3) and fps == 25.0
        w = flow_io.DenseFlowWriter(os.path.join(d, "clip.flow"))
        w.open_and_write_header(W, H, flow_io.FLOW_BACKWARD)
        for k in range(1, N):
            w.add_flow_frame(synth.const_flow(W, H))
        w.close()
        a = subprocess.run([exe, "--input_file", video, "--flow", "1"], capture_output=True, text=True,
                           timeout=300)
        assert a.returncode == 0, a.stderr
        assert "frames=45 first_frame_regions=240 total_regions=12385 label_fnv1a32=5ef008e2" in a.stdout
        # without the .flow file the same clip is segmented without flow (different pin: no-flow run)
        os.remove(os.path.join(d, "clip.flow"))
        b = subprocess.run([exe, "--input_file", video, "--flow", "1"], capture_output=True, text=True,
                           timeout=300)
        assert b.returncode == 0, b.stderr
        assert "frames=45" in b.stdout and "label_fnv1a32=5ef008e2" not in b.stdout

