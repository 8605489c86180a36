#!/usr/bin/env python
"""Interleaved A/B/... measurement of the V-JEPA step inside ONE process on one GPU.

MI355X step time drifts by several per cent with the thermal state of the package and from box to box, more than
most kernel-level deltas, so two separate bench.py runs cannot resolve a 1-2 % change.  Here one Trainer is built once,
every arm is a set of run-time options (vj_set_option names, plus the host-side switches below), and the arms take
turns: round r runs `--steps` steps of arm 0, then arm 1, ... with a device synchronise around each block.  Reported:
per-arm median / min ms per step and the paired per-round delta against arm 0 (median, and how many rounds agree in sign).

    python tools/abab.py --arms "base;4w:gemm_4w=1;tn:wgrad_tn=1" --rounds 6 --steps 6
host-side switches:  no_overlap=1 (single stream), overlap_fwd=0 (target forward on the main stream), upd_overlap=0 (the fused
                     AdamW / EMA update on the main stream, as rounds 1-4 ran it), ln_fold=1 (the target encoder's LayerNorms folded into its qkv / fc1 GEMMs),
                     tgt_flags=F (GEMM selection of the target encoder: flags | first block << 16, e.g. 786688 = 0x100 from block 12),
                     pred_flags=F / pred_dgrad_flags=F (the same for the predictor's forward / backward chain, e.g. 256 = the 4-wave kernel)
"""
import argparse
import json
import os
import statistics
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "6")
import torch  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import bench  # noqa: E402


def parse_arms(spec):
    arms = []
    for a in spec.split(";"):
        a = a.strip()
        if not a:
            continue
        name, _, rest = a.partition(":")
        opts = {}
        for kv in rest.split(","):
            if kv.strip():
                k, _, v = kv.partition("=")
                opts[k.strip()] = int(v)
        arms.append((name, opts))
    return arms


HOST_SWITCHES = ("no_overlap", "overlap_fwd", "tgt_flags", "pred_flags", "pred_dgrad_flags", "upd_overlap", "upd_prio", "ln_fold")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--arms", required=True)
    ap.add_argument("--rounds", type=int, default=6)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="vitl16")
    ap.add_argument("--out", default=None, help="write the JSON result here")
    ap.add_argument("--power", action="store_true", help="sample package power / sclk per block (tools/power.py)")
    args = ap.parse_args()
    from jepa_amd.engine import step as step_mod
    from jepa_amd.engine.layers import side_stream
    from jepa_amd.hip.lib import get_option, set_option

    device = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    wl = dict(bench.WORKLOADS[args.workload])
    trainer, sched, wd_sched = bench.build(wl, device, 1)
    batches = bench.make_inputs(wl, 8, 0, device)
    side = side_stream(device)
    arms = parse_arms(args.arms)
    lib_opts = sorted({k for _, o in arms for k in o if k not in HOST_SWITCHES})
    defaults = {k: get_option(k) for k in lib_opts}
    sampler = None
    if args.power:
        from power import PowerSampler
        sampler = PowerSampler(period=0.1)

    def apply(opts):
        for k in lib_opts:
            set_option(k, opts.get(k, defaults[k]))
        side.enabled = not opts.get("no_overlap", 0)
        step_mod._OVERLAP_FWD = bool(opts.get("overlap_fwd", 1))
        trainer.sync_update()
        trainer.overlap_update = bool(opts.get("upd_overlap", 1))
        fold = bool(opts.get("ln_fold", 0))                        # 1: the target encoder's LayerNorms folded into its GEMMs
        if fold != trainer.ln_fold_target:
            torch.cuda.synchronize()
            trainer.set_ln_fold(fold)
        lowp = bool(opts.get("upd_prio", 0))                        # 1: the update stream at the device's lowest priority
        if lowp != getattr(trainer, "_upd_lowp", False):
            torch.cuda.synchronize()
            step_mod._UPD_LOW_PRIO, trainer._upd_stream, trainer._upd_lowp = lowp, None, lowp   # fused update on its own stream, range by range (Trainer(overlap_update=))
        step_mod._TGT_GEMM_FLAGS = int(opts.get("tgt_flags", 0))   # vj_blocks_fwd gemm_flags of the target encoder (flags | first block << 16)
        step_mod._PRED_GEMM_FLAGS = int(opts.get("pred_flags", 0))        # the same for the predictor's forward chain
        step_mod._PRED_DGRAD_FLAGS = int(opts.get("pred_dgrad_flags", 0))  # option gemm_dgrad_flags around the predictor's backward chain only

    def run_steps(n, first=0):
        # every block runs the SAME batches (first, first+1, ...): mask sizes differ by batch and move the step time by
        # +-2 ms, which would otherwise alias into the arm comparison (seen in profiles/r03_abab_persist_trip2.md)
        for i in range(n):
            clips, me, mp = batches[(first + i) % len(batches)]
            trainer.train_step(clips, me, mp, lr=1e-4, wd=0.04, ema=0.998)

    # warm-up of every arm (workspace growth, kernel attribute set-up, code paging) before anything is timed
    for ai, (name, opts) in enumerate(arms):
        apply(opts)
        run_steps(len(batches) if ai == 0 else args.warmup)   # the first arm visits every batch: all workspaces reach their final size
        torch.cuda.synchronize()
    res = {name: [] for name, _ in arms}
    pw = {name: [] for name, _ in arms}
    for r in range(args.rounds):
        order = arms if r % 2 == 0 else arms[::-1]      # ABBA: cancels a linear drift inside a round
        for name, opts in order:
            apply(opts)
            run_steps(1, first=7)                         # one untimed step after the switch
            torch.cuda.synchronize()
            if sampler:
                sampler.__enter__()
            t0 = time.perf_counter()
            run_steps(args.steps)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / args.steps * 1e3
            if sampler:
                sampler.__exit__()
                pw[name].append(sampler.summary(skip_s=0.1))
            res[name].append(dt)
        print(f"[abab] round {r}: " + "  ".join(f"{n}={res[n][-1]:.2f}" for n, _ in arms), file=sys.stderr, flush=True)
    apply({})
    base = arms[0][0]
    out = {"workload": wl["desc"], "rounds": args.rounds, "steps_per_block": args.steps, "arms": {}}
    print(f"| arm | options | median ms/step | min | delta vs {base} (median of paired rounds) | rounds faster / slower | W | sclk MHz |")
    print("|---|---|---|---|---|---|---|---|")
    for name, opts in arms:
        v = res[name]
        d = [a - b for a, b in zip(v, res[base])]
        ent = {"options": opts, "ms": [round(x, 3) for x in v], "median_ms": round(statistics.median(v), 3),
               "min_ms": round(min(v), 3), "paired_delta_ms": round(statistics.median(d), 3),
               "rounds_faster": sum(1 for x in d if x < 0), "rounds_slower": sum(1 for x in d if x > 0)}
        w = s = ""
        if pw[name]:
            ws = [p.get("power_w") for p in pw[name] if p.get("power_w")]
            cs = [p.get("sclk_mhz") for p in pw[name] if p.get("sclk_mhz")]
            if ws:
                ent["power_w"] = round(sum(ws) / len(ws), 1)
                w = f"{ent['power_w']:.0f}"
            if cs:
                ent["sclk_mhz"] = round(sum(cs) / len(cs), 1)
                s = f"{ent['sclk_mhz']:.0f}"
        out["arms"][name] = ent
        pct = 100.0 * ent["paired_delta_ms"] / statistics.median(res[base])
        print(f"| {name} | {opts or '-'} | {ent['median_ms']:.2f} | {ent['min_ms']:.2f} | {ent['paired_delta_ms']:+.2f} ({pct:+.1f} %) | "
              f"{ent['rounds_faster']} / {ent['rounds_slower']} | {w} | {s} |")
    if args.out:
        with open(args.out, "w") as f:
            json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
