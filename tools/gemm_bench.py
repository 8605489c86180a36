#!/usr/bin/env python
"""GEMM micro-benchmark over the shapes of the ViT-L V-JEPA step: every tile configuration, random operands
(never zeros: DVFS), HIP events on the launch stream.  python tools/gemm_bench.py [--reps 20]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from jepa_amd.hip import ops  # noqa: E402

SHAPES = [  # (tag, M, N, K, epilogue)
    ("tgt qkv", 37632, 3072, 1024, 0), ("tgt proj", 37632, 1024, 1024, 0), ("tgt fc1", 37632, 4096, 1024, 1),
    ("tgt fc2", 37632, 1024, 4096, 0), ("tgt patch", 37632, 1024, 1536, 0),
    ("ctx qkv", 10560, 3072, 1024, 0), ("ctx proj", 10560, 1024, 1024, 0), ("ctx fc1", 10560, 4096, 1024, 1),
    ("ctx fc2", 10560, 1024, 4096, 0), ("ctx dfc2", 10560, 4096, 1024, 2), ("ctx dqkv", 10560, 1024, 3072, 0),
    ("prd qkv", 55680, 1152, 384, 0), ("prd proj", 55680, 384, 384, 0), ("prd fc1", 55680, 1536, 384, 1),
    ("prd fc2", 55680, 384, 1536, 0), ("prd dqkv", 55680, 384, 1152, 0),
    ("wg qkv", 3072, 1024, 10560, 3), ("wg proj", 1024, 1024, 10560, 3), ("wg fc1", 4096, 1024, 10560, 3),
    ("wg fc2", 1024, 4096, 10560, 3), ("wg p.qkv", 1152, 384, 27904, 3), ("wg p.proj", 384, 384, 27904, 3),
    ("wg p.fc1", 1536, 384, 27904, 3), ("wg p.fc2", 384, 1536, 27904, 3),
    ("sq 4096", 4096, 4096, 4096, 0), ("sq 8192", 8192, 8192, 8192, 0),
]
STEP_SHAPES = [s for s in SHAPES if not s[0].startswith("wg")]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--cfgs", default="1.1,2.1,2.3", help="tile.pipeline pairs (see gemm.hip dispatch_gemm)")
    ap.add_argument("--no-wgrad", action="store_true", help="skip the split-K weight-gradient shapes")
    ap.add_argument("--toggle", default=None, help="run-time option (vj_set_option name[=v0,v1,...]) measured at each value (default 0 and 1) "
                                                   "for every configuration")
    ap.add_argument("--only", default=None, help="comma-separated substrings of the shape tags to run")
    args = ap.parse_args()
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(0)
    from jepa_amd.hip.lib import set_option
    # "4.0" = flags 0x100 (gemm4w.hip); "8.x" = automatic selection with the persistent 256x256 kernel (gemm8p.hip) enabled
    cfgs = [tuple(int(v) for v in c.split(".")) + (None,) for c in args.cfgs.split(",")]
    tvals = (0, 1)
    if args.toggle and "=" in args.toggle:
        args.toggle, _, tv = args.toggle.partition("=")
        tvals = tuple(int(v) for v in tv.split(","))
    if args.toggle:
        cfgs = [(c, q, t) for c, q, _ in cfgs for t in tvals]
    print(f"{'shape':10s} {'M':>6s} {'N':>5s} {'K':>6s} epi " +
          " ".join(f"t{c}p{q}" + (f"{args.toggle}={t}" if t is not None else "") + "(TF/s)" for c, q, t in cfgs))
    shapes = STEP_SHAPES if args.no_wgrad else SHAPES
    if args.only:
        shapes = [s for s in shapes if any(k.strip() in s[0] for k in args.only.split(","))]
    for tag, M, N, K, epi in shapes:
        A = torch.randn(M, K, device=dev, generator=g).to(torch.bfloat16)
        B = (torch.randn(N, K, device=dev, generator=g) * 0.05).to(torch.bfloat16)
        bias = torch.randn(N, device=dev, generator=g)
        aux = torch.randn(M, N, device=dev, generator=g).to(torch.bfloat16) if epi in (1, 2) else None
        out = torch.empty(M, N, device=dev, dtype=torch.float32 if epi == 3 else torch.bfloat16)
        res = []
        for c, q, tg in cfgs:
            flags = 0x100 if c == 4 else (0 if c == 8 else (c << 4) | (q << 6))
            set_option("gemm_persist", 2 if c == 8 else 0)
            if tg is not None:
                set_option(args.toggle, tg)      # (after the line above: --cfgs 8.0 --toggle gemm_persist=3,1 compares forms of the persistent kernel)

            def run():
                if epi == 3:
                    ops.gemm_wgrad(A, B, out, flags=flags)
                elif epi == 1:
                    # the target encoder runs no backward: its fc1 epilogue does not save gelu'
                    ops.gemm_nt(A, B, out=out, bias=bias, aux_out=None if tag.startswith("tgt") else aux, epilogue=1, flags=flags)
                elif epi == 2:
                    ops.gemm_nt(A, B, out=out, aux_in=aux, epilogue=2, flags=flags)
                else:
                    ops.gemm_nt(A, B, out=out, bias=bias, flags=flags)
            for _ in range(3):
                run()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(args.reps):
                run()
            e.record()
            torch.cuda.synchronize()
            ms = s.elapsed_time(e) / args.reps
            res.append(2.0 * M * N * K / ms / 1e9)
        print(f"{tag:10s} {M:6d} {N:5d} {K:6d} {epi:3d} " + " ".join(f"{r:12.1f}" for r in res), flush=True)


if __name__ == "__main__":
    main()
