"""Multi-GPU chunk chain, exercised on CPU: world_size 2 over gloo with the oracle as the engine.

Every rank segments its chunks of ONE video from a fresh engine, importing the two label planes and
the counters the previous chunk's rank sent.  The concatenation of all ranks' outputs must be byte
identical to a single stream over the whole video (the reference's own single-process behaviour).
"""
import os
import pickle
import socket
import sys
import tempfile

import numpy as np
import pytest

import oracle_lib as ol
import synth

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)

W, H, N, CHUNK = 48, 36, 40, 8


def single_stream():
    s = ol.OracleStream(W, H, ol.default_options(chunk_size=CHUNK), has_flow=True)
    fl = synth.const_flow(W, H)
    out = []
    for k in range(N):
        n = s.process_frame(synth.bench_frame(W, H, k), fl if k > 0 else None, flush=(k == N - 1))
        out += [s.result_bytes(i) for i in range(n)]
    return out


def test_chunk_plan():
    from video_segment_amd.multi_gpu import chunk_plan
    assert chunk_plan(8, 20) == [(0, 7)]
    assert chunk_plan(20, 20) == [(0, 19)]
    assert chunk_plan(21, 20) == [(0, 19), (19, 20)]
    assert chunk_plan(45, 20) == [(0, 19), (19, 38), (38, 44)]
    assert chunk_plan(39, 20) == [(0, 19), (19, 38)]


def test_chain_single_process_matches_single_stream():
    """Fresh engine per chunk + halo import/export == one continuous stream."""
    from video_segment_amd.multi_gpu import local_transport, run_chain
    fl = synth.const_flow(W, H)
    got = run_chain(lambda: ol.OracleStream(W, H, ol.default_options(chunk_size=CHUNK), has_flow=True),
                    lambda k: synth.bench_frame(W, H, k), lambda k: fl, N, CHUNK, W, H, 0, 1,
                    local_transport(W, H))
    want = single_stream()
    assert [k for k, _ in got] == list(range(N))
    assert [b for _, b in got] == want


def test_chain_halo_first_order_matches_single_stream():
    """The halo-then-frames order (vsg_stream_import_halo on a fresh stream) gives the same bytes."""
    from video_segment_amd.multi_gpu import local_transport, run_chain
    fl = synth.const_flow(W, H)
    got = run_chain(lambda: ol.OracleStream(W, H, ol.default_options(chunk_size=CHUNK), has_flow=True),
                    lambda k: synth.bench_frame(W, H, k), lambda k: fl, N, CHUNK, W, H, 0, 1,
                    local_transport(W, H), overlapped=False)
    assert [b for _, b in got] == single_stream()


def _worker(rank, world, port, outfile):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, HERE)
    import torch
    import torch.distributed as dist
    from video_segment_amd.multi_gpu import DistTransport, run_chain
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    fl = synth.const_flow(W, H)
    tr = DistTransport(torch.device("cpu"), W, H, to_labels=lambda t: t.cpu().numpy())
    got = run_chain(lambda: ol.OracleStream(W, H, ol.default_options(chunk_size=CHUNK), has_flow=True),
                    lambda k: synth.bench_frame(W, H, k), lambda k: fl, N, CHUNK, W, H, rank, world,
                    tr)
    from video_segment_amd.multi_gpu import chain_nonce
    nonces = (chain_nonce(), chain_nonce())     # what two launches in a row would draw
    with open(outfile, "wb") as f:
        pickle.dump((got, nonces), f)
    dist.barrier()
    dist.destroy_process_group()


def test_chain_world2_gloo():
    import torch.multiprocessing as mp
    ol.build_oracle()
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    tmp = tempfile.mkdtemp()
    files = [os.path.join(tmp, "r%d.pkl" % r) for r in range(2)]
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_worker, args=(r, 2, port, files[r])) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    got, nonces = [], []
    for f in files:
        with open(f, "rb") as fh:
            g, nn = pickle.load(fh)
            got += g
            nonces.append(nn)
    # the nonce of the RCCL id file (vsg_chain_create): the same on every rank of a launch, another
    # one for the next launch although the launcher's environment has not changed
    assert nonces[0] == nonces[1] and nonces[0][0] != nonces[0][1] and all(0 < n < 2 ** 63 for n in nonces[0])
    got.sort(key=lambda kv: kv[0])
    want = single_stream()
    assert [k for k, _ in got] == list(range(N))
    assert [b for _, b in got] == want
