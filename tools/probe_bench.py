#!/usr/bin/env python
"""Attentive probe on frozen features (row f4 widened): time of one training step (forward + backward of AttentiveClassifier, no
optimizer) and of the cross-attention kernels alone, at the reference's eval shape (ViT-L tokens of one 16x224x224 clip per sample).
python tools/probe_bench.py [--batch 16] [--reps 20]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from jepa_amd.hip import ops  # noqa: E402
from jepa_amd.src.models.attentive_pooler import AttentiveClassifier  # noqa: E402


def timeit(fn, reps):
    for _ in range(3):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--reps", type=int, default=20)
    a = ap.parse_args()
    dev = "cuda"
    for tag, N, D, H in (("ViT-L 16x224", 1568, 1024, 16), ("ViT-H 16x384", 4608, 1280, 16)):
        B, hd = a.batch, D // H
        torch.manual_seed(0)
        m = AttentiveClassifier(embed_dim=D, num_heads=H, depth=1, num_classes=400).to(dev)
        x = torch.randn(B, N, D, device=dev).to(torch.bfloat16)
        y = torch.randint(0, 400, (B,), device=dev)

        def step():
            for p in m.parameters():
                p.grad = None
            torch.nn.functional.cross_entropy(m(x), y).backward()
        us = timeit(step, a.reps)
        flop = 3 * 2.0 * B * N * D * 2 * D      # kv projection forward + dgrad + wgrad dominate
        print(f"{tag}: probe step B={B}: {us:8.1f} us  ({B / us * 1e6:7.0f} samples/s, kv-projection GEMMs {flop / us / 1e6:6.0f} TF/s)")
        q = torch.randn(1, D, device=dev).to(torch.bfloat16)
        kv = torch.randn(B * N, 2 * D, device=dev).to(torch.bfloat16)
        dy = torch.randn(B, D, device=dev).to(torch.bfloat16)
        out, lse = ops.xattn_fwd(q, kv, B, 1, N, H, hd, hd ** -0.5)
        uf = timeit(lambda: ops.xattn_fwd(q, kv, B, 1, N, H, hd, hd ** -0.5), a.reps)
        ub = timeit(lambda: ops.xattn_bwd(q, kv, dy, lse, B, N, H, hd, hd ** -0.5), a.reps)
        byt = B * N * 2 * D * 2.0
        print(f"{tag}: xattn fwd {uf:7.1f} us = {byt / uf / 1e6:5.2f} TB/s of K+V read once | bwd {ub:7.1f} us = {3 * byt / ub / 1e6:5.2f} TB/s "
              f"(K, V read twice, dK, dV written)")


if __name__ == "__main__":
    main()
