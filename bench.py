#!/usr/bin/env python
"""Benchmark of the MI355X dense over-segmentation hot path.

Metric (BASELINE.json): over-segmented frames/sec at 1080p.
Workload at N=1 (BASELINE.json configs[2]): 1920x1080 synthetic gradient + moving checker + noise
video with a precomputed constant backward flow, chunk_size 20, spatial + temporal dense graph,
streamed through the DenseSegmentation drop-in (vsg_stream_*) with frames and flow already
resident in HBM.

A *step* is one chunk boundary of the stream: the call that segments the buffered chunk graph and
returns chunk_size-1 = 19 SegmentationDesc (steady state; every timed step is a constrained
chunk).  Warm-up steps include the first (unconstrained) chunk.

N > 1 (one process per GPU, launched by torch.distributed.run): every rank segments its own,
independent 1080p stream (different noise seed) -- the path partitions by video; the chunks of ONE
video form a dependency chain (chunk c+1 is constrained by chunk c's labels), see DESIGN.md.
No data-path collective is needed; ranks only meet for the timing barrier and the reductions.
`--mode chain` instead shards the chunks of one video round-robin over the ranks and hands the
two label planes + counters to the next rank with RCCL send/recv (the reference-exact multi-GPU
mode; it cannot scale because of the chain).

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

# A stream handle owns three HIP streams (main + two for the ordinary workers beside the tree replay);
# the HIP runtime maps all streams of a process onto GPU_MAX_HW_QUEUES hardware queues (default 4)
# and kernels of different streams that share a queue run one after the other.  With S streams per
# process the runtime therefore needs 3 S queues (measured, 4 threads: 286 -> 348 frames/s with 16
# queues; one stream: no difference).  Read once, at the first HIP call of the process.
if "--streams" in sys.argv and "GPU_MAX_HW_QUEUES" not in os.environ:
    try:
        _s = int(sys.argv[sys.argv.index("--streams") + 1])
        if _s > 1:
            os.environ["GPU_MAX_HW_QUEUES"] = str(min(3 * _s + 1, 24))
    except (IndexError, ValueError):
        pass

import numpy as np  # noqa: E402
import torch  # noqa: E402  (first: a single HIP runtime per process, see video_segment_amd/_lib.py)
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBPS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
BYTES_PER_PX_FRAME = 111.0      # spatial + temporal + flow
# Algorithmic bytes of the two candidates for the dominant kernel (the one with the larger
# summed launch time in the timed region is reported):
# k_merge_wave, per replayed edge: edge record (sorted index 4 + two root hints 8 + kept position
#   4) + two parent words 8 + two 21-byte region states (desc_sz 16, cons 4, flags 1);
# k_spine, per side cluster absorbed: child vertex 4 + orientation 4 + one parent word read 4 +
#   one 21-byte region state + the parent word written 4.
WAVE_BYTES_PER_EDGE = 4 + 8 + 4 + 8 + 2 * 21
SPINE_BYTES_PER_EDGE = 4 + 4 + 4 + 21 + 4


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=None,
                    help="ranks (one per GPU); default: WORLD_SIZE when a launcher set it, else 1")
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--chunk", type=int, default=20)
    ap.add_argument("--mode", choices=["streams", "chain"], default="streams")
    ap.add_argument("--streams", type=int, default=1,
                    help="concurrent independent 1080p streams per GPU (default 1 = the BASELINE "
                         "workload; >1 only quantifies how idle one stream leaves the GPU)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--host-inputs", action="store_true",
                    help="frames and flow are handed over as host buffers (the H2D copies are "
                         "inside the timed region); reported as config.inputs, never the default")
    ap.add_argument("--no-pcie-leg", action="store_true",
                    help="skip the short extra leg that measures the PCIe-inclusive rate")
    ap.add_argument("--dist-backend", default="nccl",
                    help="torch.distributed backend (nccl = RCCL; gloo only to exercise the N>1 "
                         "control flow on a box with fewer GPUs than ranks, with --share-gpu)")
    ap.add_argument("--share-gpu", action="store_true",
                    help="testing only: all ranks use device 0")
    ap.add_argument("--cpu-frames", type=int, default=20)
    ap.add_argument("--pipelined-leg", action="store_true",
                    help="also measure two chunk engines on one video (video_segment_amd/pipelined.py)")
    ap.add_argument("--no-chain-leg", action="store_true",
                    help="N > 1, --mode streams: skip the short chunk-chain leg after the replicas leg")
    ap.add_argument("--chain-leg-timeout", type=int, default=240)
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the information-only legs after the timed region (other configs, other "
                         "inputs, several streams per GPU)")
    return ap.parse_args()


def cpu_baseline(W, H, chunk, n_frames, device):
    """Times the CPU oracle on the first n_frames frames of the same workload (one flushed
    chunk), outside the timed region: single threaded (the reported `cpu_baseline`) and with the
    reference's default threading (one thread per Add*Edges call of the graph construction,
    row-parallel bilateral filter; the merge is serial either way) -- SURVEY 8(d)(i)/(ii).  Since
    the oracle's output is there anyway, it is compared byte for byte with what the HIP path
    produces for the same frames ("parity_checked").  Reported baselines only."""
    import oracle_lib as ol
    import synth
    import video_segment_amd as vsg
    fl = synth.const_flow(W, H)
    frames = [synth.bench_frame(W, H, k) for k in range(n_frames)]

    def run_oracle(threads):
        ol.set_threads(threads)
        st = ol.OracleStream(W, H, ol.default_options(chunk_size=chunk), has_flow=True)
        t0_ = time.perf_counter()
        n_ = 0
        for k in range(n_frames):
            n_ += st.process_frame(frames[k], fl if k > 0 else None, flush=(k == n_frames - 1))
        return st, n_, time.perf_counter() - t0_

    cores = max(1, min(os.cpu_count() or 1, 8))
    threaded = None
    if cores > 1:
        st, n_t, dt_t = run_oracle(cores)
        st.close()
        threaded = {
            "value": n_frames / dt_t, "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": "same sample, reference default threading (parallel graph construction: one "
                      "thread per edge list; %d-way row-parallel bilateral; serial merge), %.1f s"
                      % (cores, dt_t),
        }
    s, out, dt = run_oracle(1)
    assert out == n_frames
    g = vsg.DenseSegmentation(W, H, vsg.default_options(chunk_size=chunk, device=device),
                              has_flow=True)
    got = 0
    for k in range(n_frames):
        got += g.process_frame(frames[k], fl if k > 0 else None, flush=(k == n_frames - 1))
    parity = got == out and all(g.result_bytes(i) == s.result_bytes(i) for i in range(out))
    parity = parity and bool((g.last_merge_stats() == s.last_merge_stats()).all())
    g.close()
    s.close()
    return {
        "value": n_frames / dt, "unit": "frames/s", "cores": 1, "kind": "port",
        "sample": "first %d frames of the same %dx%d workload (flow, chunk %d) as one flushed "
                  "chunk, oracle/libvs_oracle.so, %.1f s" % (n_frames, W, H, chunk, dt),
    }, threaded, parity


ACC_KEYS = ["wave_ms", "wave_launches", "wave_edges", "spine_ms", "spine_launches", "spine_edges",
            "merge_ms", "pre_ms", "edges_ms", "readout_ms", "host_ms", "filter_ms", "filter_launches",
            "edges_total", "merges"]
DIAG_KEYS = ["stages", "rollbacks", "slab_growths", "spine_pool_growths", "runtime_mallocs", "runtime_frees",
             "device_syncs", "mail_waits", "mail_wait_ms", "prepare_ms", "constrained_merge_ms"]


def add_timings(a, t):
    a["wave_ms"] += t.wave_kernel_ms
    a["wave_launches"] += t.wave_kernel_launches
    a["wave_edges"] += t.wave_kernel_edges
    a["spine_ms"] += t.spine_kernel_ms
    a["spine_launches"] += t.spine_kernel_launches
    a["spine_edges"] += t.spine_kernel_edges
    a["filter_ms"] += t.filter_kernel_ms
    a["filter_launches"] += t.filter_kernel_launches
    a["merge_ms"] += t.merge_ms
    a["pre_ms"] += t.preprocess_ms
    a["edges_ms"] += t.edges_ms
    a["readout_ms"] += t.readout_ms
    a["host_ms"] += t.host_post_ms
    a["edges_total"] += t.edges_total
    a["merges"] += t.merges


def make_frames(kind, W, H, n, dev, seed_shift=0, host=False):
    """n frames of a synthetic input, generated on the device (tests/synth.py: bit-identical to the
    numpy generators the parity tests use)."""
    import synth
    frames = [synth.frame_torch(kind, W, H, k + seed_shift, dev) for k in range(n)]
    if host:
        return [f.cpu().numpy() for f in frames]
    return frames


def time_streams(vsg, frames, flow, W, H, chunk, S, warm, steps, device_index, barrier=None):
    """S independent streams (one host thread each) over the same resident frames (flow: one resident
    field for every frame, or a list with the field of frame k): `warm` untimed
    chunk boundaries per stream, then exactly `steps` timed ones between two synchronisations.
    Every SegmentationDesc of a boundary is fetched inside the timed region."""
    import threading
    torch.cuda.synchronize()
    in_use_before = vsg.memory_stats(device_index)["bytes_in_use"]
    streams = [vsg.DenseSegmentation(W, H, vsg.default_options(chunk_size=chunk, device=device_index),
                                     has_flow=True) for _ in range(S)]
    torch.cuda.synchronize()
    accs = [dict((k_, 0) for k_ in ACC_KEYS + DIAG_KEYS) for _ in range(S)]
    step_ms = [[] for _ in range(S)]
    outs = [0] * S
    pos = [0] * S
    errors = []

    def run_stream(si, n_steps, record):
        stream = streams[si]
        done = 0
        t_last = time.perf_counter()
        while done < n_steps:
            k = pos[si]
            n = stream.process_frame(frames[k], (flow[k] if isinstance(flow, list) else flow) if k > 0 else None)
            pos[si] = k + 1
            if n:
                # the consumer side of the boundary: every SegmentationDesc is fetched
                # (serialized message copied out) inside the timed region
                fetched = sum(len(stream.result_bytes(i)) for i in range(n))
                assert fetched > 0
                done += 1
                if record:
                    outs[si] += n
                    add_timings(accs[si], stream.last_timings())
                    dg = stream.last_diagnostics()
                    for k_ in DIAG_KEYS:
                        accs[si][k_] += dg[k_]
                    now = time.perf_counter()
                    step_ms[si].append((now - t_last) * 1e3)
                    t_last = now
                else:
                    t_last = time.perf_counter()

    def run(si, n_steps, record):
        try:
            run_stream(si, n_steps, record)
        except BaseException as e:  # noqa: BLE001 -- re-raised on the main thread
            errors.append(e)

    def run_all(n_steps, record):
        if S == 1:
            run_stream(0, n_steps, record)
            return
        th = [threading.Thread(target=run, args=(si, n_steps, record)) for si in range(S)]
        for t_ in th:
            t_.start()
        for t_ in th:
            t_.join()
        if errors:
            raise errors[0]

    def sync():
        torch.cuda.synchronize()
        if barrier is not None:
            barrier()

    run_all(warm, False)          # warm-up: includes the first (unconstrained) chunk
    sync()
    t0 = time.perf_counter()
    run_all(steps, True)          # exactly `steps` timed boundaries per stream
    sync()
    dt = time.perf_counter() - t0
    # what the streams hold on the device after warm-up and the timed chunks (their scratch only
    # grows): the library's own account of the blocks its live handles hold (vsg_device_memory_stats;
    # the drop in free device memory would miss blocks adopted from the cache of closed handles)
    device_bytes = max(0, vsg.memory_stats(device_index)["bytes_in_use"] - in_use_before) // S
    for st_ in streams:
        st_.close()
    acc = dict((k_, sum(a[k_] for a in accs) / S) for k_ in ACC_KEYS + DIAG_KEYS)
    all_steps = sorted(x for l_ in step_ms for x in l_)
    spread = ({"min": all_steps[0], "median": all_steps[len(all_steps) // 2], "max": all_steps[-1]}
              if all_steps else None)
    return {"dt": dt, "frames": sum(outs), "acc": acc, "device_bytes_per_stream": int(device_bytes),
            "step_ms": spread}


def measured_copy_bandwidth(dev, nbytes=1 << 30, reps=5):
    """the roofline also against what a device-to-device copy reaches on this box
    (bytes read + bytes written per second), timed with HIP events on torch's current stream."""
    a = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    b = torch.empty_like(a)
    b.copy_(a)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        b.copy_(a)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    del a, b
    return 2.0 * nbytes / (ms * 1e-3) / 1e9


def kernel_table(W, H, chunk):
    """Per-kernel rows (name, ms per step, raw HBM GB/s, fraction of peak) of the top kernels, from
    the committed rocprofv3 summary of this same command (tools/measure_round.sh writes
    profiles/<round>_kernel_table.json): counters cannot be collected inside the timed process."""
    if (W, H, chunk) != (1920, 1080, 20):
        return None
    for tag in PROFILE_TAGS:
        path = os.path.join(ROOT, "profiles", "%s_kernel_table.json" % tag)
        if os.path.exists(path):
            t = json.load(open(path))
            if t.get("source_hash") != kernel_source_hash():
                # counters of other kernels than the ones that just ran are not quoted
                return {"stale": True, "note": "%s was taken from other kernel sources (hash %s, running %s): "
                                               "run tools/measure_round.sh" % (
                                                   os.path.basename(path), t.get("source_hash"), kernel_source_hash())}
            return t
    return None


PROFILE_TAGS = ("r6", "r5", "r4", "r3", "r2")


def kernel_source_hash():
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from source_hash import source_hash
    return source_hash(ROOT)


def stage_ms(acc, steps):
    return {"preprocess": acc["pre_ms"] / steps, "edges_sort": acc["edges_ms"] / steps,
            "merge": acc["merge_ms"] / steps, "merge_wave_kernel": acc["wave_ms"] / steps,
            "merge_spine_kernel": acc["spine_ms"] / steps, "merge_filter_kernel": acc["filter_ms"] / steps,
            "readout": acc["readout_ms"] / steps, "host_post": acc["host_ms"] / steps}


def extra_measurements(vsg, args, dev, device_index, headline_fps, out):
    """What the headline number depends on (rank 0, N = 1, after the timed region; information
    only): the other single-GPU configs of BASELINE.json, the same 1080p shape on inputs with many
    small regions, and S concurrent streams on the one GPU.  Fills `out` leg by leg (a leg that
    fails must not cost the headline line: the caller records the error)."""
    import oracle_lib as ol
    import synth
    W, H, chunk = args.width, args.height, args.chunk
    px_bytes = BYTES_PER_PX_FRAME
    out["configs"] = {}

    # ---- BASELINE configs[1]: 640x480, 32-slice window, spatial-only graph through seam 3 ----
    cw, chh, cf = 640, 480, 32
    frames = [synth.frame_torch("bench", cw, chh, k, dev) for k in range(cf)]

    phase_names = ("create", "add_frames", "segment", "read_out", "close")

    def run_graph():
        """One window on a fresh graph handle, as the reference's callers use the interface
        (dense_seg_graph_interface.h:58-98); returns (#regions, phase ms, diagnostics of segment)."""
        t = [time.perf_counter()]
        g = vsg.DenseSegGraph(cw, chh, cf, device=device_index)
        t.append(time.perf_counter())
        for f in frames:
            g.add_frame_bgr(f)
        g.finish_building()
        t.append(time.perf_counter())
        g.segment(983, False)
        t.append(time.perf_counter())
        g.obtain_results(use_flows=False)
        n = g.num_regions()
        t.append(time.perf_counter())
        diag = g.diagnostics()
        g.close()
        t.append(time.perf_counter())
        return n, {k: (t[i + 1] - t[i]) * 1e3 for i, k in enumerate(phase_names)}, diag

    mem0 = vsg.memory_stats(device_index)
    _, first_phases, first_diag = run_graph()      # the first window of the process at this size
    torch.cuda.synchronize()
    mem1 = vsg.memory_stats(device_index)
    reps = 8
    windows = []
    t0 = time.perf_counter()
    for _ in range(reps):
        tw = time.perf_counter()
        nreg, ph, diag = run_graph()
        windows.append({"ms": (time.perf_counter() - tw) * 1e3, "phases": ph, "diag": diag})
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    mem2 = vsg.memory_stats(device_index)
    wms = sorted(w["ms"] for w in windows)
    slowest = max(windows, key=lambda w: w["ms"])
    diag_keys = ("stages", "rollbacks", "slab_growths", "slab_growth_ms", "spine_pool_growths",
                 "spine_pool_growth_ms", "runtime_mallocs", "runtime_malloc_ms", "runtime_frees",
                 "runtime_free_ms", "cache_hits", "device_syncs", "device_sync_ms", "mail_waits",
                 "mail_wait_ms", "mail_wait_longest_ms", "mail_mode", "prepare_ms", "segment_wall_ms")
    c1 = {"workload": "640x480 bench generator, 32-slice window, spatial-only dense graph "
                      "(DenseSegGraphInterface seam: a fresh graph per window -- create, add frames, "
                      "SegmentFullGraph(983), ObtainResults + DetermineNeighborIds, delete), frames "
                      "resident in HBM; %d timed windows after the process's first window at this size" % reps,
          "value": cf / dt, "unit": "frames/s", "ms_per_window": dt * 1e3, "regions": nreg,
          "window_ms": {"min": wms[0], "median": wms[len(wms) // 2], "max": wms[-1],
                        "all": [round(w["ms"], 2) for w in windows]},
          "phase_ms_per_window": {k: sum(w["phases"][k] for w in windows) / reps for k in phase_names},
          "segment_diagnostics_mean": {k: sum(w["diag"][k] for w in windows) / reps for k in diag_keys},
          "slowest_window": {"ms": slowest["ms"], "phases": slowest["phases"],
                             "diag": {k: slowest["diag"][k] for k in diag_keys}},
          "first_window": {"phases": first_phases, "diag": {k: first_diag[k] for k in diag_keys},
                           "runtime_mallocs": mem1["runtime_mallocs"] - mem0["runtime_mallocs"],
                           "runtime_malloc_ms": mem1["runtime_malloc_ms"] - mem0["runtime_malloc_ms"]},
          "device_memory": {"runtime_mallocs_in_timed_windows": mem2["runtime_mallocs"] - mem1["runtime_mallocs"],
                            "runtime_frees_in_timed_windows": mem2["runtime_frees"] - mem1["runtime_frees"],
                            "cache_hits_in_timed_windows": mem2["cache_hits"] - mem1["cache_hits"],
                            "bytes_cached": mem2["bytes_cached"], "cache_limit_bytes": mem2["limit_bytes"]},
          "roofline_note": "55 B/px/frame (SURVEY 8(d), spatial-only) -> %.2f GB/s" %
                           (cf / dt * cw * chh * 55.0 / 1e9)}
    if not args.no_cpu_baseline:
        hf = [f.cpu().numpy() for f in frames]
        ol.set_threads(1)
        t0 = time.perf_counter()
        og = ol.OracleGraph(cw, chh, cf)
        for f in hf:
            og.add_frame(ol.preprocess(f))
        og.segment(983, False)
        og.obtain_results(None, True, True)
        dt_o = time.perf_counter() - t0
        c1["cpu_baseline"] = {"value": cf / dt_o, "unit": "frames/s", "cores": 1, "kind": "port",
                              "sample": "the same 32-slice window, oracle/libvs_oracle.so, %.1f s" % dt_o}
        c1["parity_checked"] = bool(og.num_regions() == nreg)
        og.close()
    out["configs"]["configs[1]"] = c1
    if os.environ.get("VSG_BENCH_EXTRAS") == "configs1":   # (debugging aid: only this leg)
        return out

    # ---- 3840x2160 + flow (the over-segmentation half of configs[4]) --------------------------
    w4, h4 = 3840, 2160
    n4 = chunk + (chunk - 1) * 2
    f4 = make_frames("bench", w4, h4, n4, dev)
    fl4 = torch.from_numpy(synth.const_flow(w4, h4)).to(dev)
    r4 = time_streams(vsg, f4, fl4, w4, h4, chunk, 1, 1, 2, device_index)
    fps4 = r4["frames"] / r4["dt"]
    c4 = {"workload": "3840x2160 bench generator + constant flow, chunk %d, one stream, 2 timed "
                      "steady-state chunks, inputs resident in HBM" % chunk,
          "value": fps4, "unit": "frames/s", "ms_per_step": r4["dt"] / 2 * 1e3,
          "stage_ms_per_step": stage_ms(r4["acc"], 2),
          "device_bytes_per_stream": r4["device_bytes_per_stream"],
          "roofline_note": "%.0f B/px/frame -> %.2f GB/s" % (px_bytes, fps4 * w4 * h4 * px_bytes / 1e9)}
    if not args.no_cpu_baseline:
        ns = 8
        hf = [f.cpu().numpy() for f in f4[:ns]]
        flh = synth.const_flow(w4, h4)
        ol.set_threads(1)
        st = ol.OracleStream(w4, h4, ol.default_options(chunk_size=chunk), has_flow=True)
        t0 = time.perf_counter()
        n_ = 0
        for k in range(ns):
            n_ += st.process_frame(hf[k], flh if k > 0 else None, flush=(k == ns - 1))
        dt_o = time.perf_counter() - t0
        st.close()
        c4["cpu_baseline"] = {"value": ns / dt_o, "unit": "frames/s", "cores": 1, "kind": "port",
                              "sample": "first %d frames of the same 3840x2160 workload as one flushed "
                                        "chunk, oracle/libvs_oracle.so, %.1f s" % (ns, dt_o)}
    del f4
    out["configs"]["3840x2160"] = c4

    # ---- BASELINE configs[4] on one GPU: 4K over-segmentation on the device + the hierarchical
    # RegionSegmentation (host) on top of it.  Input: the low-contrast variant of the generator --
    # on the bench input itself the reference's RegionAgglomerationGraph aborts (neighbouring
    # checker cells are at distance exactly 1.0, region_segmentation_graph.cpp:165).
    # five chunks: with two the stream is all pipeline fill and drain (the hierarchical unit only
    # starts when the dense one has finished its first chunk)
    nh = chunk + 4 * (chunk - 1)
    fh = make_frames("soft", w4, h4, nh, dev)
    fh_host = [f.cpu().numpy() for f in fh]
    flh = synth.const_flow(w4, h4)
    dseg = vsg.DenseSegmentation(w4, h4, vsg.default_options(chunk_size=chunk, device=device_index), has_flow=True)
    rseg = vsg.RegionSegmentation(w4, h4, vsg.default_region_options())
    torch.cuda.synchronize()
    # The two units on their own threads, as the reference runs them under --use_pipeline
    # (video_pipeline.h): the dense unit hands every serialized SegmentationDesc to the hierarchical
    # one through a queue (the ctypes calls release the GIL).
    import queue
    import threading
    q = queue.Queue(maxsize=2 * chunk)
    state = {"t_region": 0.0, "n_out": 0, "error": None}

    def region_unit():
        fed = 0
        try:
            while True:
                item = q.get()
                if item is None:
                    break
                seg, last = item
                tb = time.perf_counter()
                m = rseg.process_frame(seg, fh_host[fed], flh if fed > 0 else None, flush=last)
                state["n_out"] += sum(1 for i in range(m) if len(rseg.result_bytes(i)) > 0)
                state["t_region"] += time.perf_counter() - tb
                fed += 1
        except Exception as e:   # noqa: BLE001 -- reported by the main thread
            state["error"] = e
            while q.get() is not None:
                pass

    th = threading.Thread(target=region_unit)
    th.start()
    t_dense = 0.0
    t0 = time.perf_counter()
    for k in range(nh):
        ta = time.perf_counter()
        n = dseg.process_frame(fh[k], fl4 if k > 0 else None, flush=(k == nh - 1))
        segs = [dseg.result_bytes(i) for i in range(n)]
        t_dense += time.perf_counter() - ta
        for j, seg in enumerate(segs):
            q.put((seg, k == nh - 1 and j == len(segs) - 1))
    q.put(None)
    th.join()
    dt = time.perf_counter() - t0
    if state["error"] is not None:
        raise state["error"]
    n_out, t_region = state["n_out"], state["t_region"]
    dseg.close()
    rseg.close()
    out["configs"]["configs[4]"] = {
        "workload": "3840x2160 low-contrast bench generator + constant flow, chunk %d, %d frames: dense "
                    "over-segmentation on the GPU, hierarchical RegionSegmentation (default options: Lab "
                    "+ flow histograms, size penalizer, vectorization) on the host, each unit on its own "
                    "thread; one GPU" % (chunk, nh),
        "value": n_out / dt, "unit": "frames/s", "frames": n_out,
        "dense_ms_per_frame": t_dense / nh * 1e3, "region_ms_per_frame": t_region / nh * 1e3,
        "note": "first (unconstrained) chunk included; the hierarchy is host work by design (SURVEY 8(f) row 3)"}
    del fh, fl4
    torch.cuda.empty_cache()

    # ---- the 1080p shape on inputs with many small regions ---------------------------------------
    flow = torch.from_numpy(synth.const_flow(W, H)).to(dev)
    nfr = chunk + (chunk - 1) * 2
    wl = {"checker (headline input)": {"value": headline_fps, "unit": "frames/s"}}
    for kind, label in (("varflow", "the headline frames with a spatially varying backward flow (synth.var_flow: "
                                    "rotation + zoom changing with the frame, cells moving on their own, "
                                    "out-of-range bands)"),
                        ("blobs", "value noise: 48 px cells of random colour moving with the flow, +-3 noise"),
                        ("noise", "gradient + independent +-40 noise per pixel and channel")):
        fr = make_frames("bench" if kind == "varflow" else kind, W, H, nfr, dev)
        wflow = [synth.flow_torch("var", W, H, k, dev) for k in range(nfr)] if kind == "varflow" else flow
        r = time_streams(vsg, fr, wflow, W, H, chunk, 1, 1, 2, device_index)
        fps = r["frames"] / r["dt"]
        wl[kind] = {"input": label, "value": fps, "unit": "frames/s", "ms_per_step": r["dt"] / 2 * 1e3,
                    "merges_per_step": r["acc"]["merges"] / 2, "stage_ms_per_step": stage_ms(r["acc"], 2)}
        if not args.no_cpu_baseline:
            # a short oracle sample of THIS input at THIS size (its first chunk, flushed; the oracle
            # threaded the way the reference threads graph construction), byte for byte -- the
            # full-size checks with a constrained chunk are tests/test_gpu_configs.py
            ns = chunk
            hf = [f.cpu().numpy() for f in fr[:ns]]
            flh = synth.const_flow(W, H)
            ol.set_threads(max(1, min(os.cpu_count() or 1, 8)))
            try:
                t0 = time.perf_counter()
                st = ol.OracleStream(W, H, ol.default_options(chunk_size=chunk), has_flow=True)
                g = vsg.DenseSegmentation(W, H, vsg.default_options(chunk_size=chunk, device=device_index),
                                          has_flow=True)
                same = True
                for k in range(ns):
                    fh_k = (synth.var_flow(W, H, k) if kind == "varflow" else flh) if k > 0 else None
                    fd_k = (wflow[k] if isinstance(wflow, list) else wflow) if k > 0 else None
                    no = st.process_frame(hf[k], fh_k, flush=(k == ns - 1))
                    ng = g.process_frame(fr[k], fd_k, flush=(k == ns - 1))
                    same = same and no == ng and all(g.result_bytes(i) == st.result_bytes(i) for i in range(no))
                same = same and bool((g.last_merge_stats() == st.last_merge_stats()).all())
                g.close()
                st.close()
            finally:
                ol.set_threads(1)
            wl[kind]["parity_checked"] = bool(same)
            wl[kind]["parity_sample"] = "first %d frames as one flushed chunk against oracle/libvs_oracle.so, %.1f s" % (
                ns, time.perf_counter() - t0)
        del fr, wflow
    out["workloads"] = wl

    # ---- the same video over two chunk engines on the one GPU (video_segment_amd/pipelined.py) ------
    # (no longer part of the default extras: since round 4 it gains nothing over one stream, see
    # DESIGN 7; tests/test_gpu_pipelined.py keeps it correct, --pipelined-leg measures it)
    if args.pipelined_leg:
        fr = make_frames("bench", W, H, chunk + (chunk - 1) * 6, dev)
        pipe = vsg.PipelinedDenseSegmentation(W, H, vsg.default_options(chunk_size=chunk, device=device_index),
                                              has_flow=True)
        got = 0
        for k, f in enumerate(fr):
            got += pipe.process_frame(f, flow if k > 0 else None, flush=(k == len(fr) - 1))
        stamps = pipe.stamps
        pipe.close()
        assert got == len(fr)
        steady = (stamps[-2] - stamps[1]) / (len(stamps) - 3)   # without the first and the flushed chunk
        out["pipelined"] = {
            "workload": "the headline video, even chunks on one DenseSegmentation engine and odd chunks on a "
                        "second one (two host threads, label planes handed over on the device): an engine "
                        "builds its chunk graph while the other one merges; byte-identical output "
                        "(tests/test_gpu_pipelined.py)",
            "value": (chunk - 1) / steady, "unit": "frames/s", "ms_per_step": steady * 1e3,
            "chunks_timed": len(stamps) - 3}
        del fr

    # ---- S concurrent streams on the one GPU -------------------------------------------------------
    # One PROCESS per stream (the control flow of --gpus S with every rank on this GPU: gloo barrier
    # and reductions, no data-path collective).  Threads of one process are measured as well
    # (--streams S): they scale worse, the HIP runtime serialises the host calls of a process and
    # one stream issues about 4000 launches and 250 synchronisations per chunk.
    import socket
    import subprocess
    del flow
    torch.cuda.empty_cache()
    # (the streams below are other PROCESSES on this GPU: what this process's closed handles left in
    # the library's device cache goes back to the runtime first)
    vsg.memory_trim(device_index)

    def sub_bench(extra):
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
        for k_ in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
            env.pop(k_, None)
        cmd = [sys.executable] + extra(port) + ["--steps", "2", "--warmup", "1", "--no-extras",
                                               "--no-cpu-baseline", "--no-pcie-leg", "--width", str(W),
                                               "--height", str(H), "--chunk", str(chunk)]
        p = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
        if p.returncode != 0:
            return None
        for line in reversed(p.stdout.strip().splitlines()):
            if line.startswith("{"):
                return json.loads(line)
        return None

    sweep = []
    for S in (1, 2, 4, 8):
        entry = {"streams": S}
        if S == 1:
            d = sub_bench(lambda port: [os.path.abspath(__file__)])
        else:
            d = sub_bench(lambda port: ["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(S),
                                        "--master-addr", "127.0.0.1", "--master-port", str(port),
                                        os.path.abspath(__file__), "--gpus", str(S), "--share-gpu",
                                        "--dist-backend", "gloo"])
        if d is not None:
            entry.update({"processes": {"value": d["value"], "unit": "frames/s/GPU",
                                        "end_to_end_gbps": d["value"] * W * H * px_bytes / 1e9,
                                        "ms_per_step_per_stream": d["ms_per_step"]}})
        if S > 1:
            d = sub_bench(lambda port: [os.path.abspath(__file__), "--streams", str(S)])
            if d is not None:
                entry["threads_of_one_process"] = {"value": d["value"], "unit": "frames/s/GPU",
                                                   "ms_per_step_per_stream": d["ms_per_step"]}
        sweep.append(entry)
    out["streams_sweep"] = sweep
    return out


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    gpus_given = args.gpus is not None
    if not gpus_given:
        args.gpus = world      # `torchrun --nproc-per-node N bench.py` without --gpus
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: start one rank per GPU ourselves.
        os.execvp(sys.executable, [
            sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
            "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
            "--master-port", os.environ.get("MASTER_PORT", "29533"), os.path.abspath(__file__)]
            + sys.argv[1:])
    if gpus_given:
        assert world == args.gpus, "--gpus %d but WORLD_SIZE=%d" % (args.gpus, world)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    assert torch.cuda.is_available(), "bench.py needs a HIP device; there is no CPU fallback"
    if args.share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world,
                                    device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.dist_backend, rank=rank, world_size=world)
    import synth
    import video_segment_amd as vsg

    W, H, chunk = args.width, args.height, args.chunk
    K, Wm = args.steps, args.warmup
    dev = torch.device("cuda", local_rank)

    if args.mode == "chain":
        from video_segment_amd.multi_gpu import run_chain_bench
        result = run_chain_bench(args, rank, world, local_rank)
    else:
        # ---- independent stream per rank --------------------------------------------------
        n_frames = chunk + (chunk - 1) * (Wm + K - 1) if (Wm + K) > 0 else 0
        seed_shift = 1000 * rank
        flow_host = synth.const_flow(W, H)
        flow = flow_host if args.host_inputs else torch.from_numpy(flow_host).to(dev)
        frames = make_frames("bench", W, H, n_frames, dev, seed_shift, host=args.host_inputs)
        S = max(1, args.streams)

        def barrier():
            if world > 1:
                dist.barrier()

        r = time_streams(vsg, frames, flow, W, H, chunk, S, Wm, K, local_rank, barrier)
        tt = torch.tensor([r["dt"]], dtype=torch.float64, device=dev)
        fo = torch.tensor([r["frames"]], dtype=torch.float64, device=dev)
        ranks_info = None
        if world > 1:
            # what every rank did on its own clock, and what the collective backend itself saw: a
            # SUM of ones over the process group is the number of ranks the all-reduce (RCCL over
            # xGMI with the nccl backend) really spanned, the devices show that they are distinct
            mine = torch.tensor([r["frames"] / r["dt"], float(local_rank), 1.0], dtype=torch.float64, device=dev)
            gathered = [torch.zeros_like(mine) for _ in range(world)]
            dist.all_gather(gathered, mine)
            ones = torch.ones(1, dtype=torch.float64, device=dev)
            dist.all_reduce(ones, op=dist.ReduceOp.SUM)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dist.all_reduce(fo, op=dist.ReduceOp.SUM)
            props = torch.cuda.get_device_properties(local_rank)
            ranks_info = {
                "backend": dist.get_backend(), "world_size": dist.get_world_size(),
                "allreduce_sum_of_ones": float(ones.item()),
                "per_rank_frames_per_s": [float(g[0].item()) for g in gathered],
                "per_rank_device_index": [int(g[1].item()) for g in gathered],
                "rank0_device": "%s (%d CUs, %.0f GB)" % (props.name, props.multi_processor_count,
                                                          props.total_memory / 1e9),
                "expected": "streams are independent (no data-path collective): >= 0.95 x N x the one-GPU "
                            "value with two free host cores per GPU, per-rank values within a few per cent "
                            "of each other (DESIGN 7)"}
        result = {"dt": float(tt.item()), "frames": float(fo.item()), "acc": r["acc"],
                  "step_ms": r["step_ms"],
                  "device_bytes_per_stream": r["device_bytes_per_stream"], "ranks": ranks_info,
                  "parallelism": "%d independent 1080p stream(s) per GPU x %d GPU(s)" % (S, world)}
        # PCIe-inclusive leg (rank 0, one stream, not `value`): the same steady-state chunks with
        # frames and flow handed over as host buffers, so that the H2D copies are timed as well.
        result["pcie"] = None
        if rank == 0 and not args.host_inputs and not args.no_pcie_leg and S == 1:
            st_ = vsg.DenseSegmentation(W, H, vsg.default_options(chunk_size=chunk, device=local_rank),
                                        has_flow=True)
            need = min(chunk + (chunk - 1) * 2, n_frames)
            frames_host = [frames[k].cpu().numpy() for k in range(need)]
            k = 0
            t_start, got, steps_done = None, 0, 0
            while k < need:
                n_ = st_.process_frame(frames_host[k], flow_host if k > 0 else None)
                k += 1
                if n_:
                    if t_start is not None:
                        got += n_
                        steps_done += 1
                    torch.cuda.synchronize()
                    if t_start is None:
                        t_start = time.perf_counter()     # after the first (unconstrained) chunk
            if t_start is not None and steps_done > 0:
                dt_h = time.perf_counter() - t_start
                result["pcie"] = {"value": got / dt_h, "unit": "frames/s", "steps": steps_done,
                                  "note": "frames (6.2 MB) and flow (16.6 MB) per frame copied from "
                                          "pageable host memory inside the timed region"}
            st_.close()
        del frames
        # The other partition of SURVEY 8(e), on every N > 1 run: ONE video sharded chunk-wise over
        # the ranks, the label-plane halo handed from rank to rank through the library's own
        # ncclSend / ncclRecv (dense_segmentation.cpp:281-331 is the hand-off it replaces).  Two
        # chunks per rank after one warm-up chunk per rank; information beside `value`, which stays
        # the replicas number.
        result["chain"] = None
        chain_timed_out = False
        if world > 1 and not args.no_chain_leg:
            import copy
            from video_segment_amd.multi_gpu import run_chain_bench
            a2 = copy.copy(args)
            a2.steps, a2.warmup = 2, 1
            vsg.memory_trim(local_rank)
            # The leg runs on a thread of its own under a time limit: it is the only part of an N > 1
            # run whose RCCL path (ncclCommInitRank with rank > 0, ncclSend / ncclRecv between two
            # devices) no one-GPU box could ever exercise, and a hang there must not cost the replicas
            # number that is already measured -- on a timeout every rank gives up on its own clock,
            # rank 0 still prints the line (chain: error) and the processes leave through os._exit.
            import threading
            box = {}

            def chain_leg():
                try:
                    box["rc"] = run_chain_bench(a2, rank, world, local_rank)
                except BaseException as e:   # noqa: BLE001 -- information only
                    box["error"] = "%s: %s" % (type(e).__name__, e)

            th_ = threading.Thread(target=chain_leg, daemon=True)
            th_.start()
            th_.join(args.chain_leg_timeout)
            if th_.is_alive():
                chain_timed_out = True
                result["chain"] = {"error": "no result within %d s (the hand-off or a barrier hangs): leg abandoned"
                                            % args.chain_leg_timeout}
            elif "error" in box:
                result["chain"] = {"error": box["error"]}
            else:
                rc = box["rc"]
                one = (result["frames"] / result["dt"]) / world
                result["chain"] = {
                    "value": rc["frames"] / rc["dt"], "unit": "frames/s (whole job, ONE video)",
                    "chunks_per_rank": a2.steps, "ms_per_chunk": rc["dt"] / (a2.steps * world) * 1e3,
                    "rccl_ranks": rc["handoff"]["rccl_ranks"], "transport": rc["handoff"]["transport"],
                    "handoff_ms": rc["handoff"]["send_ms_per_chunk"],
                    "recv_wait_ms": rc["handoff"]["recv_wait_ms_per_chunk"],
                    "bytes_per_handoff": rc["handoff"]["bytes_per_handoff"],
                    "vs_one_stream": (rc["frames"] / rc["dt"]) / one if one > 0 else None,
                    "expected": "<= 1.15 x one stream whatever N: the merge of chunk c+1 needs the labels "
                                "of chunk c, only graph construction and read-out overlap (DESIGN 7)"}

    if rank == 0:
        dt, frames_total, acc = result["dt"], result["frames"], result["acc"]
        fps = frames_total / dt
        px = W * H
        # the dominant kernel of the timed region (HIP events around every launch, on the stream
        # the kernel is launched on)
        if acc["spine_ms"] >= acc["wave_ms"]:
            dom = ("spine", "vsg::k_spine (large components replayed along their Kruskal tree: side "
                   "clusters absorbed in rank order, 64 per step, f32 mean as a systolic recurrence)",
                   SPINE_BYTES_PER_EDGE, "side cluster absorbed")
        else:
            dom = ("wave", "vsg::k_merge_wave (ordered per-component union-find replay, round based)",
                   WAVE_BYTES_PER_EDGE, "replayed edge")
        dom_launches = max(acc[dom[0] + "_launches"], 1)
        wave_avg_s = acc[dom[0] + "_ms"] / 1e3 / dom_launches
        wave_bytes_per_launch = dom[2] * acc[dom[0] + "_edges"] / dom_launches
        achieved = wave_bytes_per_launch / wave_avg_s / 1e9 if wave_avg_s > 0 else 0.0
        # HBM traffic of the dominant kernel per launch: rocprofv3 PMC passes cannot run inside the
        # timed process, so the committed summary of the same workload is quoted (null if absent).
        traffic, traffic_note = None, "no PMC summary under profiles/"
        traffic_same_run = None
        pmc_path = None
        for tag in PROFILE_TAGS:
            cand = os.path.join(ROOT, "profiles", "%s_pmc_%s.json" % (tag, dom[0]))
            if os.path.exists(cand):
                pmc_path = cand
                break
        pmc = json.load(open(pmc_path)) if pmc_path else None
        if pmc is not None and pmc.get("source_hash") != kernel_source_hash():
            traffic_note = ("%s was taken from other kernel sources (hash %s, running %s): not quoted; "
                            "tools/measure_round.sh collects it" % (os.path.basename(pmc_path),
                                                                   pmc.get("source_hash"), kernel_source_hash()))
            pmc = None
        if pmc is not None and (W, H, chunk) == (1920, 1080, 20):
            traffic = pmc["fetch_bytes_per_launch_raw"] + pmc["write_bytes_per_launch_raw"]
            traffic_note = ("FETCH_SIZE + WRITE_SIZE per launch, raw counters, from " + pmc["source"])
            # (the counter passes are a run of their own, with their own number of launches: the
            # algorithmic bytes per launch OF THAT RUN are what the traffic is to be held against)
            traffic_same_run = pmc.get("algorithmic_bytes_per_launch_same_run")
        # algorithmic bytes per launch of BOTH replay kernels in this run: under the counter passes
        # the slower clock of the instrumented run can make the other kernel the dominant one
        alg_by_kernel = {
            "wave": WAVE_BYTES_PER_EDGE * acc["wave_edges"] / max(acc["wave_launches"], 1),
            "spine": SPINE_BYTES_PER_EDGE * acc["spine_edges"] / max(acc["spine_launches"], 1)}
        copy_gbps = measured_copy_bandwidth(dev) if args.mode == "streams" else None
        out = {
            "metric": "over-segmented frames/sec at 1080p",
            "value": fps,
            "unit": "frames/s",
            "n_gpus": world,
            "steps": K,
            "warmup": Wm,
            "ms_per_step": dt / K * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": "BASELINE configs[2]: %dx%d synthetic gradient + moving checker + "
                            "+-3 noise video, precomputed constant backward flow (-2,0), "
                            "spatial+temporal dense graph, chunk_size %d, DenseSegmentation "
                            "stream API, inputs resident in HBM" % (W, H, chunk),
                "frames_per_step": chunk - 1,
                "parallelism": result["parallelism"],
                "mode": args.mode,
                "inputs": "host buffers (H2D inside the timed region)" if args.host_inputs
                          else "resident in HBM",
            },
            "roofline": {
                "bound": "hbm",
                "kernel": dom[1],
                "achieved": achieved,
                "peak": HBM_PEAK_GBPS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBPS,
                "traffic": traffic,
                "traffic_unit": "B/launch",
                "traffic_note": traffic_note,
                "algorithmic_bytes_per_launch_by_kernel": alg_by_kernel,
                "traffic_vs_algorithmic_same_run": (traffic / traffic_same_run) if (traffic and traffic_same_run)
                else None,
                "peak_measured": copy_gbps,
                "peak_measured_note": "device-to-device copy of 1 GiB on this box, bytes read + written per "
                                      "second (SURVEY 8(d))",
                "frac_of_measured": (achieved / copy_gbps) if copy_gbps else None,
                "kernels": kernel_table(W, H, chunk),
                "launches": acc[dom[0] + "_launches"],
                "avg_launch_ms": wave_avg_s * 1e3,
                "bytes_per_launch": wave_bytes_per_launch,
                "note": "dependency-bound serial replay: algorithmic bytes = %d B per %s "
                        "(DESIGN.md); whole path = %.1f B/px/frame -> %.2f GB/s end to end"
                        % (dom[2], dom[3], BYTES_PER_PX_FRAME,
                           fps / world * px * BYTES_PER_PX_FRAME / 1e9),
            },
            "stage_ms_per_step": stage_ms(acc, K),
            "step_ms_spread": result.get("step_ms"),
            "diagnostics_per_step": {k_: acc[k_] / K for k_ in DIAG_KEYS if k_ in acc},
            "device_bytes_per_stream": result.get("device_bytes_per_stream"),
            "ranks": result.get("ranks"),
            "chain": result.get("chain"),
            "edges_per_step": acc["edges_total"] / K,
            "merges_per_step": acc["merges"] / K,
        }
        if result.get("pcie"):
            out["pcie_inclusive"] = result["pcie"]
        if result.get("handoff"):
            out["config"]["handoff"] = result["handoff"]
        if not args.no_cpu_baseline and world == 1:   # (rank 0 at N = 1 only: the other ranks would wait)
            out["cpu_baseline"], threaded, out["parity_checked"] = cpu_baseline(
                W, H, chunk, args.cpu_frames, local_rank)
            if threaded is not None:
                out["cpu_baseline_threaded"] = threaded
            assert out["parity_checked"], "HIP output differs from the oracle on the bench workload"
        if world == 1 and args.mode == "streams" and not args.no_extras and not args.host_inputs \
                and args.streams == 1 and (W, H) == (1920, 1080):
            torch.cuda.empty_cache()
            extras = {}
            try:
                extra_measurements(vsg, args, dev, local_rank, fps, extras)
            except Exception as e:   # noqa: BLE001 -- information only: the headline line still goes out
                extras["extras_error"] = "%s: %s" % (type(e).__name__, e)
            out.update(extras)
        print(json.dumps(out), flush=True)
    if world > 1:
        if args.mode == "streams" and chain_timed_out:
            sys.stdout.flush()
            os._exit(0)   # (a thread of this process still hangs in the abandoned leg)
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
