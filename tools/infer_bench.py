#!/usr/bin/env python
"""Frozen-encoder forward throughput (what the reference's evals run, evals/video_classification_frozen/
eval.py:414-441 loads the target encoder and calls it under no_grad): ViT-L/16 or ViT-H/16 on 16x224x224 clips.

    python tools/infer_bench.py [--model vit_large] [--batch 24] [--reps 10]
Prints clips/s for the module-level no_grad forward with the automatic GEMM selection and with the two-workgroups-per-CU
GEMM (the default for inference), and the algorithmic TFLOP/s (engine/flops.py)."""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="vit_large")
    ap.add_argument("--batch", type=int, default=24)
    ap.add_argument("--reps", type=int, default=10)
    args = ap.parse_args()
    import jepa_amd.src.models.vision_transformer as V
    from jepa_amd.engine.flops import block_flops
    torch.manual_seed(0)
    vit = V.__dict__[args.model](img_size=224, patch_size=16, num_frames=16, tubelet_size=2, uniform_power=True).cuda()
    for p in vit.parameters():
        p.requires_grad = False
    clips = torch.randn(args.batch, 3, 16, 224, 224, device="cuda")
    N, D, depth = vit.num_patches, vit.embed_dim, len(vit.blocks)
    flop = args.batch * (depth * block_flops(N, D) + 2 * N * 1536 * D)
    for name, flags in (("auto (8-phase 256x256)", 0), ("two workgroups per CU (256x128)", 0x100)):
        V.INFER_GEMM_FLAGS = flags
        with torch.no_grad():
            for _ in range(2):
                y = vit(clips)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.reps):
                y = vit(clips)
            torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / args.reps
        print(f"{args.model} B={args.batch} {name:34s}: {1e3 * dt:7.2f} ms  {args.batch / dt:7.1f} clips/s  "
              f"{flop / dt / 1e12:6.1f} TFLOP/s  (out {tuple(y.shape)}, {y.dtype})", flush=True)


if __name__ == "__main__":
    main()
