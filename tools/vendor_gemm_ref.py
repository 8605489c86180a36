#!/usr/bin/env python
"""Reference point only (never on the product path): the vendor GEMM PyTorch-ROCm dispatches to (hipBLASLt / rocBLAS behind
F.linear) on the step's forward shapes, next to our kernels, same box, random operands, HIP events.  Shows how much of the
distance to the MFMA roof is the kernel and how much is the shape / the chip's power-limited clock.

    python tools/vendor_gemm_ref.py [--reps 20]"""
import argparse
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from jepa_amd.hip import ops  # noqa: E402

SHAPES = [("tgt qkv", 37632, 3072, 1024), ("tgt proj", 37632, 1024, 1024), ("tgt fc1 (no gelu)", 37632, 4096, 1024),
          ("tgt fc2", 37632, 1024, 4096), ("ctx qkv", 10560, 3072, 1024), ("ctx proj", 10560, 1024, 1024),
          ("ctx fc2", 10560, 1024, 4096), ("prd qkv", 58560, 1152, 384), ("prd proj", 58560, 384, 384),
          ("prd fc2", 58560, 384, 1536), ("vith qkv", 37632, 3840, 1280), ("vith fc2", 37632, 1280, 5120),
          ("sq 4096", 4096, 4096, 4096), ("sq 8192", 8192, 8192, 8192)]


def timeit(fn, reps):
    for _ in range(3):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=20)
    args = ap.parse_args()
    g = torch.Generator(device="cuda").manual_seed(0)
    print(f"{'shape':18s} {'M':>6s} {'N':>5s} {'K':>5s} | ours 8-phase | ours 4-wave | vendor (F.linear) TF/s")
    for tag, M, N, K in SHAPES:
        A = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
        W = (torch.randn(N, K, device="cuda", generator=g) * 0.05).to(torch.bfloat16)
        b32 = torch.randn(N, device="cuda", generator=g)
        b16 = b32.to(torch.bfloat16)
        out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        fl = 2.0 * M * N * K
        t8 = timeit(lambda: ops.gemm_nt(A, W, out=out, bias=b32, flags=0xC0), args.reps)
        t4 = timeit(lambda: ops.gemm_nt(A, W, out=out, bias=b32, flags=0x100), args.reps)
        tv = timeit(lambda: F.linear(A, W, b16), args.reps)
        print(f"{tag:18s} {M:6d} {N:5d} {K:5d} | {fl / t8 / 1e9:10.0f}   | {fl / t4 / 1e9:9.0f}   | {fl / tv / 1e9:9.0f}", flush=True)


if __name__ == "__main__":
    main()
