"""bench.py contract: one JSON line with the required keys; the N > 1 control flow (barriers,
reductions, chunk-chain hand-off) exercised with two ranks sharing the one GPU of the test box
(gloo instead of RCCL, which refuses two ranks on one device)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REQUIRED = ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
            "scaling", "vs_baseline", "dtype", "data", "config", "roofline"]


def last_json_line(text):
    for line in reversed(text.strip().splitlines()):
        if line.startswith("{"):
            return json.loads(line)
    raise AssertionError(text)


def check(out, n_gpus, steps, warmup):
    for k in REQUIRED:
        assert k in out, k
    assert out["n_gpus"] == n_gpus and out["steps"] == steps and out["warmup"] == warmup
    assert out["value"] > 0 and out["higher_is_better"] is True and out["scaling"] == "weak"
    assert out["unit"] == "frames/s" and out["dtype"] == "f32" and out["data"] == "synthetic"
    assert "workload" in out["config"] and "model" not in out["config"]
    r = out["roofline"]
    for k in ["bound", "achieved", "peak", "unit", "frac", "traffic"]:
        assert k in r, k
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12


@pytest.mark.gpu
def test_single_gpu_line_and_cpu_baseline():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1",
                        "--width", "320", "--height", "240", "--cpu-frames", "20"],
                       capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    out = last_json_line(p.stdout)
    check(out, 1, 2, 1)
    c = out["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] == 1 and c["value"] > 0 and "sample" in c


@pytest.mark.gpu
@pytest.mark.parametrize("mode,port", [("streams", "29611"), ("chain", "29612")])
def test_two_ranks_control_flow(mode, port):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", port, os.path.join(ROOT, "bench.py"),
           "--gpus", "2", "--steps", "2", "--warmup", "1", "--width", "256", "--height", "144",
           "--mode", mode, "--dist-backend", "gloo", "--share-gpu", "--no-cpu-baseline"]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=360, cwd=ROOT, env=env)
    assert p.returncode == 0, (p.stdout + p.stderr)[-3000:]
    out = last_json_line(p.stdout)
    check(out, 2, 2, 1)
    # two ranks x two timed steps x 19 frames (streams) or one 4-chunk video (chain: 20 + 3 x 19)
    frames = out["value"] * out["ms_per_step"] * out["steps"] / 1e3
    assert abs(frames - (76 if mode == "streams" else 77)) < 1e-6
    if mode == "streams":
        # the N > 1 line says by itself how many ranks the collective backend spanned and what
        # every one of them measured
        rk = out["ranks"]
        assert rk["world_size"] == 2 and rk["allreduce_sum_of_ones"] == 2.0 and rk["backend"] == "gloo"
        assert len(rk["per_rank_frames_per_s"]) == 2 and all(v > 0 for v in rk["per_rank_frames_per_s"])
        assert out["value"] <= sum(rk["per_rank_frames_per_s"]) * (1 + 1e-9)   # MAX over ranks of the time
        # ... and carries the other partition of SURVEY 8(e) as well: a short chunk-chain leg (ONE
        # video over the ranks, the halo handed on through the transport: RCCL inside the library
        # on a multi-GPU node, torch.distributed/gloo in this one-GPU control-flow test)
        ch = out["chain"]
        assert "error" not in ch, ch
        assert ch["value"] > 0 and ch["chunks_per_rank"] == 2 and ch["rccl_ranks"] == 0
        assert ch["handoff_ms"] >= 0 and ch["recv_wait_ms"] >= 0 and ch["bytes_per_handoff"] == 2 * 256 * 144 * 4 + 32
        assert "expected" in ch and ch["vs_one_stream"] > 0


def test_committed_round_line_carries_every_key():
    """The default `python bench.py` line of the round (profiles/, produced on an MI355X by
    tools/measure_round.sh): the contract keys plus what the headline depends on -- the other
    single-GPU configs, the other inputs (each with its own oracle check), the streams sweep, the
    memory a stream holds -- and counters that belong to the kernel sources of the same commit."""
    import glob
    paths = sorted(glob.glob(os.path.join(ROOT, "profiles", "r6_*_bench_1080p.json")))
    assert paths, "no round-6 bench line under profiles/"
    out = last_json_line(open(paths[-1]).read())
    check(out, 1, out["steps"], out["warmup"])
    assert out["vs_baseline"] is None and out["parity_checked"] is True
    c = out["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] == 1 and c["value"] > 0
    assert out["roofline"]["bound"] == "hbm" and out["roofline"]["peak"] == 8000.0
    cfg = out["configs"]
    assert cfg["configs[1]"]["cpu_baseline"]["value"] > 0 and cfg["configs[1]"]["parity_checked"] is True
    assert set(cfg["configs[1]"]["phase_ms_per_window"]) == {"create", "add_frames", "segment", "read_out", "close"}
    assert cfg["3840x2160"]["value"] > 0 and cfg["configs[4]"]["value"] > 0
    # four 3840x2160 streams fit the 288 GB of the device
    assert (1 << 30) < cfg["3840x2160"]["device_bytes_per_stream"] < 72e9
    wl = out["workloads"]
    assert set(wl) >= {"checker (headline input)", "blobs", "noise"}
    assert wl["blobs"]["parity_checked"] is True and wl["noise"]["parity_checked"] is True
    assert [s["streams"] for s in out["streams_sweep"]] == [1, 2, 4, 8]
    assert "pipelined" not in out          # (only with --pipelined-leg)
    r = out["roofline"]
    assert r["peak_measured"] > 1000 and 0 < r["frac_of_measured"] < 1
    assert len(r["kernels"]["top"]) == 8 and all(k["ms_per_step"] > 0 for k in r["kernels"]["top"])
    assert 0.9 < r["traffic_vs_algorithmic_same_run"] < 1.5
    assert (1 << 30) < out["device_bytes_per_stream"] < 24e9
    # the counters the line quotes were taken from the kernel sources that are committed with it
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from source_hash import source_hash
    assert r["kernels"]["source_hash"] == source_hash(ROOT)
    for name in ("r6_pmc_wave.json", "r6_pmc_spine.json", "r6_kernel_table.json"):
        assert json.load(open(os.path.join(ROOT, "profiles", name)))["source_hash"] == source_hash(ROOT), name
