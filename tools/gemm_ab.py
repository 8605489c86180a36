#!/usr/bin/env python
"""Interleaved A/B of one run-time option on single GEMM shapes (HIP events, warm chip): the values take turns, `--rounds` times each,
and the median per value is printed.  A first-measured configuration on a cool chip reads 10 - 20 % high (tools/gemm_bench.py measures
its configurations one after the other), which is what this tool avoids.
    python tools/gemm_ab.py --option gemm_persist --values 3,1 --shapes 55680x1152x384,55680x384x384:res"""
import argparse
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from jepa_amd.hip import ops  # noqa: E402
from jepa_amd.hip.lib import set_option  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--option", required=True)
    ap.add_argument("--values", required=True)
    ap.add_argument("--shapes", required=True, help="comma list of MxNxK[:res|:gelu|:dgelu]")
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--reps", type=int, default=30)
    args = ap.parse_args()
    vals = [int(v) for v in args.values.split(",")]
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(0)
    print(f"| shape | epilogue | " + " | ".join(f"{args.option}={v}: us (TF/s)" for v in vals) + " | last / first |")
    print("|---|---|" + "---|" * (len(vals) + 1))
    for spec in args.shapes.split(","):
        dims, _, kind = spec.partition(":")
        M, N, K = (int(x) for x in dims.split("x"))
        A = torch.randn(M, K, device=dev, generator=g).to(torch.bfloat16)
        W = (torch.randn(N, K, device=dev, generator=g) * 0.05).to(torch.bfloat16)
        bias = torch.randn(N, device=dev, generator=g)
        other = torch.randn(M, N, device=dev, generator=g).to(torch.bfloat16)
        out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)

        def run():
            if kind == "res":
                ops.gemm_nt(A, W, out=out, bias=bias, residual=other)
            elif kind == "gelu":
                ops.gemm_nt(A, W, out=out, bias=bias, aux_out=other, epilogue=ops.EPI_GELU)
            elif kind == "dgelu":
                ops.gemm_nt(A, W, out=out, aux_in=other, epilogue=ops.EPI_DGELU)
            else:
                ops.gemm_nt(A, W, out=out, bias=bias)
        for _ in range(200):      # warm the chip on this shape
            run()
        torch.cuda.synchronize()
        t = {v: [] for v in vals}
        for _ in range(args.rounds):
            for v in vals:
                old = set_option(args.option, v)
                for _ in range(5):
                    run()
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                for _ in range(args.reps):
                    run()
                e.record()
                torch.cuda.synchronize()
                set_option(args.option, old)
                t[v].append(s.elapsed_time(e) / args.reps * 1e3)
        med = [statistics.median(t[v]) for v in vals]
        print(f"| {M} x {N} x {K} | {kind or 'bias'} | " + " | ".join(f"{m:.1f} ({2.0 * M * N * K / m / 1e6:.0f})" for m in med) +
              f" | {med[-1] / med[0]:.3f} |", flush=True)


if __name__ == "__main__":
    main()
