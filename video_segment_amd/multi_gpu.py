"""Multi-GPU chunk chain: one video, chunks round-robin over the ranks.

The reference is a single process: chunk c+1 is constrained by the labels chunk c produced for
its two overlap frames, and region ids continue from `max_region_id_`.  Sharding one video over
several GPUs therefore is a *pipeline*: rank r = c mod world segments chunk c and hands
    * the region-id image of the virtual overlap frame      (W*H int32)
    * the region-id image of the constrained overlap frame  (W*H int32)
    * {max_region_id, chunk_id, num_output_frames, input_frames}
to the rank that owns chunk c+1 (point-to-point send/recv: RCCL over xGMI on GPUs; gloo in the
CPU tests).  Graph *construction* needs no exchange: every rank re-filters the raw frames of its
own chunk, including the constrained overlap frame.

The runner is engine agnostic: the engine factory returns an object with the DenseSegmentation
interface (process_frame / result_bytes / export_halo / import_halo); the product engine is
video_segment_amd.DenseSegmentation, the CPU tests use the oracle.
"""
import numpy as np


def chunk_plan(num_frames, chunk):
    """Frames fed to the engine of every chunk: list of (first_frame, last_frame) inclusive.

    Chunk 0 is fed frames 0..chunk-1; chunk c >= 1 is fed its constrained overlap frame
    s = c*(chunk-1) first and then s+1 .. s+chunk-1 (dense_segmentation.cpp:157, 281-331: the
    steady state advances chunk-1 output frames per chunk).  The last chunk ends with the video.
    """
    plan = []
    first = 0
    stride = chunk - 1
    while True:
        last = min(first + chunk - 1, num_frames - 1)
        plan.append((first, last))
        if last >= num_frames - 1:
            break
        first += stride
    return plan


class DistTransport:
    """torch.distributed point-to-point transport (backend nccl = RCCL on GPUs, gloo on CPU).

    halo_of(engine) -> (labels_virtual, labels_constrained, scalars): what is sent (default:
    engine.export_halo(), host arrays -- the oracle engine of the CPU tests; for the product engine
    pass lambda e: product_halo(e, W, H, dev)).  to_labels(x): converts a received plane to what
    engine.import_halo accepts."""
    name = "torch.distributed send/recv"

    def __init__(self, device, width, height, halo_of=None, to_labels=None):
        import torch
        import torch.distributed as dist
        self.torch, self.dist, self.device = torch, dist, device
        self.W, self.H = width, height
        self.halo_of, self.to_labels = halo_of, to_labels
        self._local = None

    def _halo(self, engine):
        halo = self.halo_of(engine) if self.halo_of is not None else engine.export_halo()
        if isinstance(halo[0], int):
            # DenseSegmentation.export_halo() hands out library-owned device pointers that die
            # with the engine's next chunk: they have to be copied first (product_halo does).
            raise TypeError("product engine: pass halo_of=lambda e: product_halo(e, W, H, dev)")
        return halo

    def send_halo(self, engine, dst, rank):
        halo = self._halo(engine)
        if dst == rank:
            self._local = halo
            return
        for a in halo:
            t = a if self.torch.is_tensor(a) else self.torch.from_numpy(np.ascontiguousarray(a))
            self.dist.send(t.to(self.device).contiguous(), dst=dst)

    def recv_halo(self, engine, src, rank):
        torch = self.torch
        if src == rank:
            virt, cons, scal = self._local
        else:
            out = []
            for shape, dtype in (((self.H, self.W), torch.int32), ((self.H, self.W), torch.int32),
                                 ((4,), torch.int64)):
                t = torch.empty(shape, dtype=dtype, device=self.device)
                self.dist.recv(t, src=src)
                out.append(t)
            virt, cons, scal = out
        if self.to_labels is not None:
            virt, cons = self.to_labels(virt), self.to_labels(cons)
        scal_np = scal.cpu().numpy() if torch.is_tensor(scal) else np.asarray(scal)
        engine.import_halo(virt, cons, scal_np)


class ChainTransport:
    """The hand-off inside the library: vsg_chain_send_halo / vsg_chain_recv_halo (ncclSend /
    ncclRecv on the chain's own stream, include/vsg.h) -- what a C++ host uses; product engine
    only.  A rank that owns two consecutive chunks (world == 1, or the tail) keeps a copy of the
    halo planes on its device and imports it without a transfer."""
    name = "vsg_chain (ncclSend / ncclRecv inside libvsg_hip)"

    def __init__(self, chain, width, height, device):
        self.chain, self.W, self.H, self.device = chain, width, height, device
        self._local = None

    def send_halo(self, engine, dst, rank):
        if dst == rank:
            self._local = product_halo(engine, self.W, self.H, self.device)
        else:
            self.chain.send_halo(engine, dst)

    def recv_halo(self, engine, src, rank):
        if src == rank:
            virt, cons, scal = self._local
            engine.import_halo(virt, cons, scal.cpu().numpy())
        else:
            self.chain.recv_halo(engine, src)


def local_transport(width, height, device=None):
    """Transport of a single process (world 1): the halo stays on the rank.  device: a torch cuda
    device for the product engine (its library-owned planes are copied), None for host engines."""
    if device is None:
        return DistTransport(None, width, height)
    return DistTransport(device, width, height,
                         halo_of=lambda e: product_halo(e, width, height, device))


def run_chain(engine_factory, get_frame, get_flow, num_frames, chunk, width, height, rank, world,
              transport, overlapped=True):
    """Segments the chunks owned by `rank` and returns [(frame_index, SegmentationDesc bytes)].

    engine_factory(): the rank's engine (has_flow must match get_flow); ONE engine serves all the
    rank's chunks (engine.restart() between them).
    get_frame(k), get_flow(k): inputs of global frame k in the engine's memory kind (flow(0) unused).
    transport: DistTransport or ChainTransport (send_halo(engine, dst, rank) / recv_halo(engine,
    src, rank)).
    overlapped: the SURVEY 8(e) order -- the rank feeds the frames of its chunk first (features,
    edges and the bucket sort do not depend on the previous chunk: all ranks build concurrently)
    and only then blocks in the receive of the halo, right before the frame that completes the
    chunk.  False: receive first (the halo-then-frames order of vsg_stream_import_halo).
    """
    plan = chunk_plan(num_frames, chunk)
    out = []
    eng = None
    for c, (first, last) in enumerate(plan):
        if c % world != rank:
            continue
        if eng is None:
            eng = engine_factory()
        else:
            eng.restart()
        if c > 0:
            if overlapped:
                eng.expect_halo()
            else:
                transport.recv_halo(eng, (c - 1) % world, rank)
        next_frame_out = None
        for k in range(first, last + 1):
            if c > 0 and overlapped and k == last:
                # the merge needs the labels; everything before it did not
                transport.recv_halo(eng, (c - 1) % world, rank)
            flush = (k == num_frames - 1)
            flow = get_flow(k) if (get_flow is not None and k > 0) else None
            n = eng.process_frame(get_frame(k), flow, flush=flush)
            if n:
                # the chunk's outputs start at frame `first` (chunk 0: frame 0)
                base = first if next_frame_out is None else next_frame_out
                for i in range(n):
                    out.append((base + i, eng.result_bytes(i)))
                next_frame_out = base + n
        if c + 1 < len(plan):
            transport.send_halo(eng, (c + 1) % world, rank)
    if eng is not None:
        eng.close()
    return out


def chain_nonce():
    """A nonce all ranks of ONE launch share and no other launch has (see vsg_chain_create: a stale
    id file of a crashed job must not be mistaken for this job's).  With a process group: rank 0
    draws it at random and broadcasts it -- the launcher's environment is not enough, a default
    torchrun start has the same TORCHELASTIC_RUN_ID ('none'), address and port every time.  Without
    one (or with VSG_CHAIN_NONCE set, for launchers of their own): derived from the environment."""
    import hashlib
    import os
    import secrets
    if "VSG_CHAIN_NONCE" not in os.environ:
        try:
            import torch.distributed as dist
            if dist.is_available() and dist.is_initialized():
                box = [secrets.randbits(63) | 1 if dist.get_rank() == 0 else None]
                dist.broadcast_object_list(box, src=0)
                return int(box[0])
        except ImportError:
            pass
    key = "|".join(os.environ.get(k, "") for k in ("VSG_CHAIN_NONCE", "TORCHELASTIC_RUN_ID", "MASTER_ADDR",
                                                    "MASTER_PORT"))
    return int.from_bytes(hashlib.sha256(key.encode()).digest()[:8], "little") >> 1


import sys


def run_chain_bench(args, rank, world, local_rank):
    """bench.py --mode chain: one long video sharded chunk-wise over the ranks; the halo travels
    through the library's own RCCL hand-off (vsg_chain_send_halo / vsg_chain_recv_halo).

    The chain is a pipeline (rank r cannot start chunk c before rank r-1 has finished chunk c-1),
    so a barrier in the middle of a video would deadlock against the blocking hand-off.  Warm-up
    and measurement are therefore two separate videos: `warmup` chunks per rank, barrier, then
    `steps` chunks per rank timed between two barriers (weak scaling: the video grows with the
    number of ranks)."""
    import os
    import tempfile
    import time
    import torch
    import torch.distributed as dist
    import video_segment_amd as vsg
    import synth

    W, H, chunk = args.width, args.height, args.chunk
    K, Wm = args.steps, args.warmup
    dev = torch.device("cuda", local_rank)
    flow = torch.from_numpy(synth.const_flow(W, H)).to(dev)
    # every rank of the launch derives the same path and nonce from the launcher's environment
    chain = None
    if args.dist_backend == "nccl" and not args.share_gpu:
        nonce = chain_nonce()
        id_file = os.path.join(tempfile.gettempdir(), "vsg_chain_%016x.id" % nonce)
        chain = vsg.ChunkChain(rank, world, id_file, nonce=nonce, device=local_rank)
        rccl_rank, rccl_world = chain.info()
        print("[bench chain] rank %d of %d: RCCL communicator rank %d of %d on device %d" %
              (rank, world, rccl_rank, rccl_world, local_rank), file=sys.stderr, flush=True)
        assert (rccl_rank, rccl_world) == (rank, world), (rccl_rank, rccl_world, rank, world)
        transport = ChainTransport(chain, W, H, dev)
    else:
        # testing only (--share-gpu / gloo: RCCL refuses two ranks on one device)
        rccl_world = 0
        transport = DistTransport(dev, W, H, halo_of=lambda e: product_halo(e, W, H, dev))
    acc = {"wave_ms": 0.0, "wave_launches": 0, "wave_edges": 0, "spine_ms": 0.0, "spine_launches": 0,
           "spine_edges": 0, "merge_ms": 0.0, "pre_ms": 0.0,
           "edges_ms": 0.0, "readout_ms": 0.0, "host_ms": 0.0, "filter_ms": 0.0,
           "filter_launches": 0, "edges_total": 0, "merges": 0}

    def load(total_chunks):
        num_frames = chunk + (chunk - 1) * (total_chunks - 1)
        plan = chunk_plan(num_frames, chunk)
        mine = [c for c in range(len(plan)) if c % world == rank]
        frames = {}
        for c in mine:
            for k in range(plan[c][0], plan[c][1] + 1):
                if k not in frames:
                    frames[k] = torch.from_numpy(synth.bench_frame(W, H, k)).to(dev)
        return num_frames, plan, mine, frames

    engine = vsg.DenseSegmentation(W, H, vsg.default_options(chunk_size=chunk, device=local_rank),
                                   has_flow=True)

    handoff = {"send_ms": 0.0, "sends": 0, "recv_ms": 0.0, "recvs": 0}

    def run_video(video, record):
        num_frames, plan, mine, frames = video
        frames_out = 0
        for c in mine:
            first, last = plan[c]
            eng = engine
            eng.restart()
            if c > 0:
                eng.expect_halo()   # build first, receive the labels right before the merge
            for k in range(first, last + 1):
                if c > 0 and k == last:
                    th = time.perf_counter()
                    transport.recv_halo(eng, (c - 1) % world, rank)
                    if record:
                        handoff["recv_ms"] += (time.perf_counter() - th) * 1e3
                        handoff["recvs"] += 1
                n = eng.process_frame(frames[k], flow if k > 0 else None, flush=(k == num_frames - 1))
                if n:
                    fetched = sum(len(eng.result_bytes(i)) for i in range(n))   # consumer side
                    assert fetched > 0
                if n and record:
                    frames_out += n
                    t = eng.last_timings()
                    acc["wave_ms"] += t.wave_kernel_ms
                    acc["wave_launches"] += t.wave_kernel_launches
                    acc["wave_edges"] += t.wave_kernel_edges
                    acc["spine_ms"] += t.spine_kernel_ms
                    acc["spine_launches"] += t.spine_kernel_launches
                    acc["spine_edges"] += t.spine_kernel_edges
                    acc["filter_ms"] += t.filter_kernel_ms
                    acc["filter_launches"] += t.filter_kernel_launches
                    acc["merge_ms"] += t.merge_ms
                    acc["pre_ms"] += t.preprocess_ms
                    acc["edges_ms"] += t.edges_ms
                    acc["readout_ms"] += t.readout_ms
                    acc["host_ms"] += t.host_post_ms
                    acc["edges_total"] += t.edges_total
                    acc["merges"] += t.merges
            if c + 1 < len(plan):
                th = time.perf_counter()
                transport.send_halo(eng, (c + 1) % world, rank)
                if record:
                    handoff["send_ms"] += (time.perf_counter() - th) * 1e3
                    handoff["sends"] += 1
        return frames_out

    warm = load(Wm * world) if Wm > 0 else None
    timed = load(K * world)
    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    if warm is not None:
        run_video(warm, False)
    barrier()
    t0 = time.perf_counter()
    frames_out = run_video(timed, True)
    barrier()
    dt = time.perf_counter() - t0
    tt = torch.tensor([dt], dtype=torch.float64, device=dev)
    fo = torch.tensor([frames_out], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dist.all_reduce(fo, op=dist.ReduceOp.SUM)
    engine.close()
    if chain is not None:
        chain.close()
    # (rank 0's own clock: a send returns when the receiver has taken the planes -- it is usually
    # waiting for them already --, a receive includes the wait for the previous rank's merge)
    return {"dt": float(tt.item()), "frames": float(fo.item()), "acc": acc,
            "handoff": {"transport": transport.name, "rccl_ranks": rccl_world,
                        "send_ms_per_chunk": handoff["send_ms"] / max(handoff["sends"], 1),
                        "recv_wait_ms_per_chunk": handoff["recv_ms"] / max(handoff["recvs"], 1),
                        "bytes_per_handoff": 2 * W * H * 4 + 32},
            "parallelism": "chain: ONE video, chunks round-robin over %d GPUs, every rank builds its "
                           "chunk graph before it blocks in the receive of the label-plane halo "
                           "(vsg_chain_send_halo / vsg_chain_recv_halo: ncclSend / ncclRecv inside "
                           "the library; the RCCL communicator reports %d ranks); a pipeline, not "
                           "data parallel" % (world, rccl_world)}


def _wrap_device_int32(ptr, n, dev):
    """Views library-owned device memory as a torch tensor (no copy) via __cuda_array_interface__."""
    import torch

    class _Holder:
        pass

    h = _Holder()
    h.__cuda_array_interface__ = {"shape": (n,), "typestr": "<i4", "data": (int(ptr), False),
                                  "version": 2}
    return torch.as_tensor(h, device=dev)


def product_halo(engine, width, height, device):
    """(labels_virtual, labels_constrained, scalars) of a product engine as torch device tensors
    (copies of the library-owned planes), ready for transport.send / import_halo."""
    import torch
    pa, pb, scal = engine.export_halo()
    n = width * height
    ta = _wrap_device_int32(pa, n, device).view(height, width).clone()
    tb = _wrap_device_int32(pb, n, device).view(height, width).clone()
    return ta, tb, torch.from_numpy(scal)
