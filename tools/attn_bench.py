#!/usr/bin/env python
"""Attention micro-benchmark on the ViT-L V-JEPA step shapes (random bf16 qkv, HIP events).
python tools/attn_bench.py [--reps 10] [--only-fwd]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from jepa_amd.hip import ops  # noqa: E402

SHAPES = [("tgt", 24, 1568, 16, 64), ("ctx m0", 24, 366, 16, 64), ("ctx m1", 24, 107, 16, 64),
          ("prd m0", 24, 1113, 16, 24), ("prd m1", 24, 1208, 16, 24), ("vith", 8, 1568, 16, 80),
          ("vith384", 2, 4608, 16, 80)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--only-fwd", action="store_true")
    ap.add_argument("--shapes", default="")
    ap.add_argument("--fwd-qt", type=int, default=2)
    args = ap.parse_args()
    dev = "cuda"
    from jepa_amd.hip.lib import load_library
    load_library().vj_attn_set_variant(args.fwd_qt)
    g = torch.Generator(device=dev).manual_seed(0)
    for tag, B, S, H, hd in SHAPES:
        if args.shapes and tag.split()[0] not in args.shapes:
            continue
        qkv = torch.randn(B * S, 3 * H * hd, device=dev, generator=g).to(torch.bfloat16)
        dout = torch.randn(B * S, H * hd, device=dev, generator=g).to(torch.bfloat16)
        scale = hd ** -0.5
        o, lse = ops.attn_fwd(qkv, B, S, H, hd, scale)

        def timeit(fn):
            for _ in range(2):
                fn()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(args.reps):
                fn()
            e.record()
            torch.cuda.synchronize()
            return s.elapsed_time(e) / args.reps
        ms_f = timeit(lambda: ops.attn_fwd(qkv, B, S, H, hd, scale))
        fl = 4.0 * B * H * S * S * hd
        line = f"{tag:7s} B{B} S{S} H{H} hd{hd}: fwd {ms_f * 1e3:8.1f} us {fl / ms_f / 1e9:7.1f} TF/s"
        if not args.only_fwd:
            ms_b = timeit(lambda: ops.attn_bwd(qkv, o, dout, lse, B, S, H, hd, scale))
            line += f" | bwd {ms_b * 1e3:8.1f} us {2 * fl / ms_b / 1e9:7.1f} TF/s"
        print(line, flush=True)


if __name__ == "__main__":
    main()
