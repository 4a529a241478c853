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
    from video_segment_amd.multi_gpu import local_transport, run_chain
    want = _stream(vsg, W, H, N, chunk)
    fl = synth.const_flow(W, H)
    dev = torch.device("cuda", 0)
    got = run_chain(
        lambda: vsg.DenseSegmentation(W, H, vsg.default_options(chunk_size=chunk), has_flow=True),
        lambda k: synth.bench_frame(W, H, k), lambda k: fl, N, chunk, W, H, 0, 1,
        local_transport(W, H, dev), overlapped=overlapped)
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
    """vsg_chain_create / info / destroy (ncclCommInitRank through the id file).  A stale record under
    the same name (another run's nonce, or garbage) is replaced by rank 0 and gone afterwards; a
    lone send to the own rank is rejected; a stream on another device than the chain too."""
    from video_segment_amd._lib import VsgError
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "id")
        with open(path, "wb") as f:
            f.write(b"stale" * 40)
        c = vsg.ChunkChain(0, 1, path, nonce=0x1234)
        assert c.info() == (0, 1)
        assert not os.path.exists(path)      # removed once the communicator exists
        s = vsg.DenseSegmentation(64, 48, vsg.default_options(chunk_size=8), has_flow=False)
        with pytest.raises(VsgError):
            c.send_halo(s, 0)
        with pytest.raises(VsgError):
            c.recv_halo(s, 0)
        with pytest.raises(VsgError):
            c.exchange_halo(s, 0, s, 0)      # a stream cannot hand the halo to itself
        s.close()
        c.close()


@pytest.mark.parametrize("W,H,N,chunk", [(96, 64, 44, 10), (320, 240, 50, 20)])
def test_rccl_loopback_handoff_matches_stream(vsg, W, H, N, chunk):
    """The library's own RCCL hand-off moving real halos on ONE GPU: consecutive chunks alternate
    between two handles of the same device and every hand-off is one vsg_chain_exchange_halo --
    ncclSend + ncclRecv of the rank to itself in one group -- instead of a device copy.  The
    result is the continuous stream, byte for byte (overlapped order: frames before the halo)."""
    from video_segment_amd.multi_gpu import chunk_plan
    want = _stream(vsg, W, H, N, chunk)
    fl = synth.const_flow(W, H)
    plan = chunk_plan(N, chunk)
    assert len(plan) >= 3
    with tempfile.TemporaryDirectory() as d:
        chain = vsg.ChunkChain(0, 1, os.path.join(d, "id"), nonce=7)
        eng = [vsg.DenseSegmentation(W, H, vsg.default_options(chunk_size=chunk), has_flow=True)
               for _ in range(2)]
        got = []
        for c, (first, last) in enumerate(plan):
            e = eng[c % 2]
            e.restart()
            if c > 0:
                e.expect_halo()
            for k in range(first, last + 1):
                if c > 0 and k == last:
                    chain.exchange_halo(eng[(c - 1) % 2], 0, e, 0)
                n = e.process_frame(synth.bench_frame(W, H, k), fl if k > 0 else None,
                                    flush=(k == N - 1))
                got += [e.result_bytes(i) for i in range(n)]
        for e in eng:
            e.close()
        chain.close()
    assert got == want


def _rccl_rank(rank, world, id_file, nonce, W, H, N, chunk, outfile):
    """One rank of the 2-GPU chain (own process, own GPU): ChainTransport = vsg_chain_send_halo /
    vsg_chain_recv_halo between the processes."""
    import pickle
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, here)
    sys.path.insert(0, os.path.dirname(here))
    import torch
    import video_segment_amd as vsg
    from video_segment_amd.multi_gpu import ChainTransport, run_chain
    torch.cuda.set_device(rank)
    dev = torch.device("cuda", rank)
    chain = vsg.ChunkChain(rank, world, id_file, nonce=nonce, device=rank)
    assert chain.info() == (rank, world)
    fl = synth.const_flow(W, H)
    got = run_chain(
        lambda: vsg.DenseSegmentation(W, H, vsg.default_options(chunk_size=chunk, device=rank), has_flow=True),
        lambda k: synth.bench_frame(W, H, k), lambda k: fl, N, chunk, W, H, rank, world,
        ChainTransport(chain, W, H, dev))
    chain.close()
    with open(outfile, "wb") as f:
        pickle.dump(got, f)


def test_rccl_two_rank_chain_matches_stream(vsg):
    """Two processes, two GPUs, the halo over RCCL (ncclSend / ncclRecv inside the library) -- needs
    two devices: RCCL refuses two ranks on one GPU, so on a one-GPU box this is skipped and
    test_rccl_loopback_handoff_matches_stream is what moves the bytes."""
    import pickle
    import torch.multiprocessing as mp
    from video_segment_amd import _lib
    if _lib.lib().vsg_device_count() < 2:
        pytest.skip("needs two GPUs (RCCL: one rank per device)")
    W, H, N, chunk = 320, 240, 90, 20
    want = _stream(vsg, W, H, N, chunk)
    tmp = tempfile.mkdtemp()
    files = [os.path.join(tmp, "r%d.pkl" % r) for r in range(2)]
    nonce = int.from_bytes(os.urandom(8), "little")
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_rccl_rank,
                         args=(r, 2, os.path.join(tmp, "id"), nonce, W, H, N, chunk, files[r]))
             for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(600)
        if p.is_alive():
            p.kill()
            pytest.fail("a rank of the RCCL chain hung")
        assert p.exitcode == 0
    got = []
    for f in files:
        with open(f, "rb") as fh:
            got += pickle.load(fh)
    got.sort(key=lambda kv: kv[0])
    assert [k for k, _ in got] == list(range(N))
    assert [b for _, b in got] == want
