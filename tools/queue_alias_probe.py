#!/usr/bin/env python
"""Hardware-queue aliasing of HIP streams and what it costs the two-stream step (round 5).

ROCclr maps streams onto GPU_MAX_HW_QUEUES hardware queues; two streams on one queue serialise.  Part 1 creates N torch streams and
prints, for each, whether an idle one-wave kernel on it runs concurrently with one on the default stream and with one on the first
created stream (vj_probe_spin, engine/layers.py streams_concurrent): the pattern shows the mapping period.  Part 2 runs the ViT-L
B=24 step with the engine's SIDE stream (target forward, weight gradients) replaced by (a) a stream that is independent of the
main stream and (b) the first stream found that aliases it, interleaved, and prints ms per step: the cost of a bad mapping --
the size of the unexplained one-rank slowdown of the `vj_comm_*` route in round 4 (profiles/r04_dp1_capi_trace.md).

    python tools/queue_alias_probe.py [--streams 24] [--rounds 3] [--steps 6] [--no-step]
"""
import argparse
import os
import statistics
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "6")
import torch  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--streams", type=int, default=24)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--no-step", action="store_true")
    args = ap.parse_args()
    from jepa_amd.engine import layers
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    main_s = torch.cuda.current_stream()
    print(f"GPU_MAX_HW_QUEUES={os.environ.get('GPU_MAX_HW_QUEUES')}")
    pool = [torch.cuda.Stream(device=dev) for _ in range(args.streams)]
    print("| stream # (creation order) | concurrent with the default stream | concurrent with stream 0 |")
    print("|---|---|---|")
    vs_main, vs_first = [], []
    for i, s in enumerate(pool):
        a = layers.streams_concurrent(s, main_s)
        b = layers.streams_concurrent(s, pool[0]) if i else False
        vs_main.append(a)
        vs_first.append(b)
        print(f"| {i} | {'yes' if a else 'NO (same queue)'} | {'-' if i == 0 else ('yes' if b else 'NO (same queue)')} |")
    print(f"\nstreams sharing the default stream's queue: {[i for i, a in enumerate(vs_main) if not a]}; sharing stream 0's queue: "
          f"{[i for i, b in enumerate(vs_first) if i and not b]}")
    if args.no_step:
        return
    good = next((s for s, a in zip(pool, vs_main) if a), None)
    bad = next((s for s, a in zip(pool, vs_main) if not a), None)
    if good is None or bad is None:
        print("no aliasing / no independent stream among the candidates: part 2 skipped")
        return
    wl = dict(bench.WORKLOADS["vitl16"])
    trainer, _, _ = bench.build(wl, dev, 1)
    trainer.overlap_update = False          # only the two streams of rounds 1-4: main + side
    batches = bench.make_inputs(wl, 8, 0, dev)
    side = layers.side_stream(dev)

    def run(n, first=0):
        for i in range(n):
            clips, me, mp = batches[(first + i) % len(batches)]
            trainer.train_step(clips, me, mp, lr=1e-4, wd=0.04, ema=0.998)
    res = {"independent": [], "aliased": []}
    for name, st in (("independent", good), ("aliased", bad)):
        torch.cuda.synchronize()
        side.stream = st
        run(len(batches))
    for r in range(args.rounds):
        for name, st in ((("independent", good), ("aliased", bad)) if r % 2 == 0 else (("aliased", bad), ("independent", good))):
            torch.cuda.synchronize()
            side.stream = st
            run(1)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            run(args.steps)
            torch.cuda.synchronize()
            res[name].append((time.perf_counter() - t0) * 1e3 / args.steps)
    print("\n| side stream | ms per step (median of rounds) | rounds |")
    print("|---|---|---|")
    for name in res:
        print(f"| {name} of the main stream's hardware queue | {statistics.median(res[name]):.2f} | {[round(x, 2) for x in res[name]]} |")


if __name__ == "__main__":
    main()
