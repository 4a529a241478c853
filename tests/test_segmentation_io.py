"""`.pb` container: Python writer/reader round trip on oracle output, and
the C++ SegmentationWriterUnit of the host mirror writing the HIP path's output (GPU)."""
import hashlib
import os
import subprocess
import tempfile

import pytest

import oracle_lib as ol
import synth
from video_segment_amd.segmentation_io import SegmentationWriter, read_segmentation_file

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "video_segment_amd", "host")


def oracle_frames(W, H, N, flow=True, chunk=20):
    s = ol.OracleStream(W, H, ol.default_options(chunk_size=chunk), has_flow=flow)
    fl = synth.const_flow(W, H) if flow else None
    out = []
    for k in range(N):
        n = s.process_frame(synth.probe_frame(W, H, k), fl if (flow and k > 0) else None,
                            flush=(k == N - 1))
        out += [s.result_bytes(i) for i in range(n)]
    return out


def write_py(path, frames, chunk_every=None):
    w = SegmentationWriter(path)
    assert w.open_file([1, 0])
    for i, f in enumerate(frames):
        w.add_segmentation_data_to_chunk(f, i * 40000)
        if chunk_every and (i + 1) % chunk_every == 0:
            w.write_chunk()
    w.write_term_header_and_close()


def test_container_round_trip():
    frames = oracle_frames(64, 48, 25)
    with tempfile.TemporaryDirectory() as d:
        p = os.path.join(d, "a.pb")
        write_py(p, frames)                       # one chunk (what the reference's unit writes)
        flags, got, chunks = read_segmentation_file(p)
        assert flags == [1, 0] and chunks == 1
        assert [b for _, b in got] == frames and [t for t, _ in got] == [i * 40000 for i in range(25)]
        write_py(p, frames, chunk_every=10)       # several chunks
        flags, got, chunks = read_segmentation_file(p)
        assert chunks == 3 and [b for _, b in got] == frames


def test_cpp_reader_reads_python_written_container():
    """SegmentationReader of the host mirror (no GPU involved): frames, time stamps, resolution
    and the label hash of the App. B pin come back from a file written by the Python writer."""
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "video_segment_amd", "csrc"), "-j8", "-s"])
    subprocess.check_call(["make", "-C", HOST, "-s"])
    W, H, N = 64, 48, 45
    frames = oracle_frames(W, H, N)
    with tempfile.TemporaryDirectory() as d:
        for chunk_every in (None, 7):
            p = os.path.join(d, "o.pb")
            write_py(p, frames, chunk_every=chunk_every)
            r = subprocess.run([os.path.join(HOST, "seg_tree_synth"), "--read_pb", p],
                               capture_output=True, text=True, timeout=120)
            assert r.returncode == 0, r.stderr
            assert "frames=45 first_frame_regions=240 total_regions=12385 label_fnv1a32=5ef008e2" in r.stdout
            assert "width=64 height=48 header_flags=2 last_pts=%d" % (44 * 40000) in r.stdout
            assert "bytes=%d " % sum(len(f) for f in frames) in r.stdout
        # truncated file: no TERM header
        q = os.path.join(d, "t.pb")
        open(q, "wb").write(open(p, "rb").read()[:-8])
        r = subprocess.run([os.path.join(HOST, "seg_tree_synth"), "--read_pb", q],
                           capture_output=True, text=True, timeout=120)
        assert r.returncode == 1


@pytest.mark.gpu
def test_host_writer_unit_matches_python_writer():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "video_segment_amd", "csrc"), "-j8", "-s"])
    subprocess.check_call(["make", "-C", HOST, "-s"])
    W, H, N = 64, 48, 45
    with tempfile.TemporaryDirectory() as d:
        p = os.path.join(d, "hip.pb")
        r = subprocess.run([os.path.join(HOST, "seg_tree_synth"), "--width", str(W), "--height", str(H),
                            "--frames", str(N), "--flow", "1", "--output_file", p],
                           capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr
        assert "label_fnv1a32=5ef008e2" in r.stdout
        flags, got, chunks = read_segmentation_file(p)
        assert flags == [1, 0] and chunks == 1 and len(got) == N
        q = os.path.join(d, "oracle.pb")
        write_py(q, oracle_frames(W, H, N))
        assert hashlib.sha256(open(p, "rb").read()).hexdigest() == \
            hashlib.sha256(open(q, "rb").read()).hexdigest()
