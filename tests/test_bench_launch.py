"""bench.py's launch contract on CPU: `python bench.py --gpus N` (N > 1, no launcher environment) must start its own N
ranks under torch.distributed.run, keep working when an external launcher already started them, report the MAX over
ranks, and have rank 0 print exactly one JSON line.  The step is replaced by bench.py's `--stub-step-ms` rehearsal (gloo,
sleep + one all-reduce); everything around it -- argument handling, self-launch, rendezvous on 127.0.0.1, fences, timing,
report -- is the code the GPU run uses (reference launcher: app/main.py:28-71, one process per device)."""
import json
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env():
    env = {k: v for k, v in os.environ.items()
           if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RUN_ID")}
    env["OMP_NUM_THREADS"] = "1"
    return env


def _json_lines(stdout):
    return [json.loads(ln) for ln in stdout.splitlines() if ln.startswith("{")]


def test_self_launch_two_ranks():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "3", "--warmup", "1", "--stub-step-ms", "20"],
                       capture_output=True, text=True, timeout=240, env=_env())
    assert r.returncode == 0, r.stderr[-2000:]
    assert "self-launch:" in r.stderr and "--nproc-per-node=2" in r.stderr
    lines = _json_lines(r.stdout)
    assert len(lines) == 1, r.stdout          # rank 0 only
    line = lines[0]
    assert line["n_gpus"] == 2 and line["steps"] == 3 and line["warmup"] == 1 and line["scaling"] == "weak"
    assert line["config"]["parallelism"] == "dp2" and line["config"]["global_batch"] == 2 * line["config"]["per_gpu_batch"]
    # rank 1 sleeps 1.5 x 20 ms per step: the reported time is the slowest rank's, not rank 0's
    assert line["ms_per_step"] >= 29.0, line
    assert abs(line["value"] - line["config"]["global_batch"] / (line["ms_per_step"] * 1e-3)) < 0.01 * line["value"]
    # the data-parallel object: per-rank step-time spread (tells mask-draw skew from communication), the route the buckets took and
    # the outcome of the collective-stream check (round 6)
    dp = line["dp"]
    for key in ("ranks", "route", "collectives", "collective_stream_check", "rank_ms_per_step", "rank_exposed_comm_ms",
                "rank_compute_ms_per_step"):
        assert key in dp, dp
    sp = dp["rank_ms_per_step"]
    assert sp["min"] <= sp["mean"] <= sp["max"] and abs(sp["max"] - line["ms_per_step"]) < 0.5
    # the all-reduce equalises the ranks' step times; their OWN work (step time minus the wait in the collective) shows that rank 1
    # sleeps 1.5 x as long as rank 0, and the fast rank's wait is about the difference
    cp, ex = dp["rank_compute_ms_per_step"], dp["rank_exposed_comm_ms"]
    assert cp["max"] >= 1.3 * cp["min"], dp
    assert ex["max"] >= 5.0 and ex["min"] <= 0.5 * ex["max"], dp


def test_external_launcher_still_works():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), BENCH, "--gpus", "2", "--steps", "2", "--warmup", "1", "--stub-step-ms", "5"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=240, env=_env())
    assert r.returncode == 0, r.stderr[-2000:]
    assert "self-launch:" not in r.stderr     # WORLD_SIZE was set by the launcher: no second level of processes
    lines = _json_lines(r.stdout)
    assert len(lines) == 1 and lines[0]["n_gpus"] == 2


def test_single_rank_line_and_missing_devices():
    r = subprocess.run([sys.executable, BENCH, "--steps", "2", "--warmup", "1", "--stub-step-ms", "5"],
                       capture_output=True, text=True, timeout=120, env=_env())
    assert r.returncode == 0, r.stderr[-2000:]
    lines = _json_lines(r.stdout)
    assert len(lines) == 1 and lines[0]["n_gpus"] == 1 and lines[0]["config"]["parallelism"] == "dp1"
    assert "self-launch" not in r.stderr
    # the real (non-stub) path with more GPUs than the node has: fails on the missing device, not on the launch form
    import torch
    if torch.cuda.device_count() < 2:
        r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "1", "--warmup", "0"], capture_output=True,
                           text=True, timeout=120, env=_env())
        assert r.returncode != 0 and "not present" in (r.stderr + r.stdout)
