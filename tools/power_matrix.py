#!/usr/bin/env python
"""Power / clock matrix of the V-JEPA step and of its kernels on one MI355X (diagnostics).

One process, one Trainer; each case loops for `--seconds` with package power and shader clock sampled at 10 Hz
(tools/power.py) and reports achieved TF/s (or GB/s), W, MHz and TF/s per kW.  Cases:
  idle | step (two streams) | step (one stream) | target-encoder forward only | our 8-phase GEMM 8192^3 and on two step
  shapes | the vendor GEMM (torch.matmul -> hipBLASLt; a reference point, never on the product path) on the same shapes |
  attention forward hd 64 | AdamW+EMA (HBM-bound) | zero-filled operands of the same GEMM (DVFS give-back check)
"""
import argparse
import json
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "6")
import torch  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import bench  # noqa: E402
from power import PowerSampler  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=8.0)
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    from jepa_amd.engine.flops import step_flops
    from jepa_amd.engine.layers import side_stream
    from jepa_amd.hip import ops
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    wl = dict(bench.WORKLOADS["vitl16"])
    trainer, _, _ = bench.build(wl, dev, 1)
    batches = bench.make_inputs(wl, 4, 0, dev)
    side = side_stream(dev)
    sampler = PowerSampler(period=0.1)
    g = torch.Generator(device=dev).manual_seed(0)
    kpe = 3 * wl["tubelet"] * wl["patch"] ** 2
    N = (wl["frames"] // wl["tubelet"]) * (wl["crop"] // wl["patch"]) ** 2
    B = wl["batch"]
    cnt = [0]

    def step():
        clips, me, mp = batches[cnt[0] % len(batches)]
        cnt[0] += 1
        trainer.train_step(clips, me, mp, lr=1e-4, wd=0.04, ema=0.998)
        return step_flops(wl["embed_dim"], wl["depth"], wl["pred_dim"], wl["pred_depth"], N, kpe, B,
                          [m.shape[1] for m in me], [m.shape[1] for m in mp])

    def tgt_fwd():
        clips, me, mp = batches[0]
        trainer.forward_target(clips, mp)
        D, L = wl["embed_dim"], wl["depth"]
        return B * (L * (24 * N * D * D + 4 * N * N * D) + 2 * N * kpe * D)

    def gemm_case(M, Nn, K, vendor=False, zero=False, epi=0):
        A = torch.randn(M, K, device=dev, generator=g).to(torch.bfloat16)
        W = (torch.randn(Nn, K, device=dev, generator=g) * 0.05).to(torch.bfloat16)
        if zero:
            A.zero_()
            W.zero_()
        out = torch.empty(M, Nn, device=dev, dtype=torch.bfloat16)
        bias = torch.zeros(Nn, device=dev)
        if vendor:
            Wt = W.t()

            def f():
                torch.matmul(A, Wt, out=out)
                return 2.0 * M * Nn * K
        else:
            def f():
                ops.gemm_nt(A, W, out=out, bias=bias, epilogue=epi)
                return 2.0 * M * Nn * K
        return f

    def attn_case(Bq, S, H, hd):
        qkv = torch.randn(Bq * S, 3 * H * hd, device=dev, generator=g).to(torch.bfloat16)
        o = torch.empty(Bq * S, H * hd, device=dev, dtype=torch.bfloat16)

        def f():
            ops.attn_fwd(qkv, Bq, S, H, hd, hd ** -0.5, save_lse=False, out=o)
            return 4.0 * Bq * H * S * S * hd
        return f

    def adamw_case():
        def f():
            trainer.optimizer_step(1e-4, 0.04, 0.998)
            return 0.0
        return f

    cases = [
        ("idle", None, {}),
        ("step, two streams (default)", step, {}),
        ("step, one stream (VJ_NO_OVERLAP)", step, {"serial": True}),
        ("step, two streams (repeat)", step, {}),
        ("target-encoder forward only", tgt_fwd, {}),
        ("our 8-phase GEMM 8192^3", gemm_case(8192, 8192, 8192), {}),
        ("vendor GEMM 8192^3 (torch.matmul)", gemm_case(8192, 8192, 8192, vendor=True), {}),
        ("our 8-phase GEMM 8192^3, ZERO operands", gemm_case(8192, 8192, 8192, zero=True), {}),
        ("our GEMM tgt qkv 37632x3072x1024", gemm_case(37632, 3072, 1024), {}),
        ("vendor GEMM tgt qkv 37632x3072x1024", gemm_case(37632, 3072, 1024, vendor=True), {}),
        ("our GEMM tgt fc1+GELU 37632x4096x1024", gemm_case(37632, 4096, 1024, epi=1), {}),
        ("our GEMM tgt fc2 37632x1024x4096", gemm_case(37632, 1024, 4096), {}),
        ("attention fwd B24 S1568 H16 hd64", attn_case(24, 1568, 16, 64), {}),
        ("AdamW+EMA+recast (HBM-bound)", adamw_case(), {}),
    ]
    rows = []
    for name, fn, kw in cases:
        side.enabled = not kw.get("serial", False)
        if fn is None:
            torch.cuda.synchronize()
            with sampler:
                time.sleep(3.0)
            s = sampler.summary(skip_s=0.5)
            rows.append(dict(case=name, tflops=0.0, **s))
            print(json.dumps(rows[-1]), file=sys.stderr, flush=True)
            continue
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        flop, n = 0.0, 0
        with sampler:
            t0 = time.perf_counter()
            while time.perf_counter() - t0 < args.seconds:
                for _ in range(4):
                    flop += fn()
                    n += 1
                torch.cuda.synchronize()
            dt = time.perf_counter() - t0
        s = sampler.summary(skip_s=1.0)
        tf = flop / dt / 1e12
        rows.append(dict(case=name, tflops=round(tf, 1), calls=n, ms_per_call=round(1e3 * dt / n, 3), **s))
        print(json.dumps(rows[-1]), file=sys.stderr, flush=True)
    side.enabled = True
    print("| case | TF/s | ms / call | package W (mean) | max W | sclk MHz (mean) | min MHz | TF/s per kW |")
    print("|---|---|---|---|---|---|---|---|")
    for r in rows:
        w = r.get("power_w")
        eff = f"{r['tflops'] / w * 1e3:.0f}" if w and r["tflops"] else ""
        print(f"| {r['case']} | {r['tflops'] or ''} | {r.get('ms_per_call', '')} | {w or ''} | {r.get('power_max_w', '')} | "
              f"{r.get('sclk_mhz', '')} | {r.get('sclk_mhz_min', '')} | {eff} |")
    if args.out:
        with open(args.out, "w") as f:
            json.dump(rows, f, indent=1)


if __name__ == "__main__":
    main()
