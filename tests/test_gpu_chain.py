"""Chunk chain with the product engine on one device: ONE handle per rank (vsg_stream_restart),
frames before the halo (vsg_stream_expect_halo: the chunk graph is built while the previous chunk
is still being segmented elsewhere) and halo first -- both byte-identical to the continuous
stream -- plus the error contract of the deferred halo and the RCCL communicator of one rank."""
import hashlib
import os
import tempfile

import numpy as np
import pytest

import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def vsg():
    import video_segment_amd as v
    from video_segment_amd import _lib
    _lib.build()
    assert _lib.lib().vsg_device_count() > 0
    return v


def _stream(vsg, W, H, N, chunk):
    s = vsg.DenseSegmentation(W, H, vsg.default_options(chunk_size=chunk), has_flow=True)
    fl = synth.const_flow(W, H)
    out = []
    for k in range(N):
        n = s.process_frame(synth.bench_frame(W, H, k), fl if k > 0 else None, flush=(k == N - 1))
        out += [s.result_bytes(i) for i in range(n)]
    s.close()
    return out


@pytest.mark.parametrize("overlapped", [True, False])
@pytest.mark.parametrize("W,H,N,chunk", [(96, 64, 44, 10), (160, 120, 50, 20)])
def test_chain_orders_match_stream(vsg, overlapped, W, H, N, chunk):
    import torch
    from video_segment_amd.multi_gpu import product_halo, run_chain
    want = _stream(vsg, W, H, N, chunk)
    fl = synth.const_flow(W, H)
    dev = torch.device("cuda", 0)
    got = run_chain(
        lambda: vsg.DenseSegmentation(W, H, vsg.default_options(chunk_size=chunk), has_flow=True),
        lambda k: synth.bench_frame(W, H, k), lambda k: fl, N, chunk, W, H, 0, 1, None,
        from_engine_halo=lambda e: product_halo(e, W, H, dev), overlapped=overlapped)
    assert [k for k, _ in got] == list(range(N))
    assert [b for _, b in got] == want


def test_deferred_halo_contract(vsg):
    from video_segment_amd._lib import VsgError
    W, H, chunk = 64, 48, 8
    fl = synth.const_flow(W, H)
    s = vsg.DenseSegmentation(W, H, vsg.default_options(chunk_size=chunk), has_flow=True)
    s.expect_halo()
    with pytest.raises(VsgError):
        s.expect_halo()                      # only on a fresh stream
    for k in range(chunk - 1):
        assert s.process_frame(synth.bench_frame(W, H, 7 + k), fl) == 0
    with pytest.raises(VsgError):            # the chunk cannot be segmented without the labels
        s.process_frame(synth.bench_frame(W, H, 7 + chunk - 1), fl)
    s.restart()                               # and the handle is usable again afterwards
    n = 0
    for k in range(chunk):
        n += s.process_frame(synth.bench_frame(W, H, k), fl if k > 0 else None, flush=(k == chunk - 1))
    assert n == chunk
    s.close()


def test_rccl_communicator_of_one_rank(vsg):
    """vsg_chain_create / destroy (ncclCommInitRank through the id file); a rank cannot send to
    itself."""
    from video_segment_amd._lib import VsgError
    with tempfile.TemporaryDirectory() as d:
        c = vsg.ChunkChain(0, 1, os.path.join(d, "id"))
        s = vsg.DenseSegmentation(64, 48, vsg.default_options(chunk_size=8), has_flow=False)
        with pytest.raises(VsgError):
            c.send_halo(s, 0)
        s.close()
        c.close()
