#!/usr/bin/env python
"""Reference point only (never on the product path): PyTorch-ROCm's F.scaled_dot_product_attention (the flash / efficient
backend it ships) forward + backward on the step's attention shapes next to our kernels, same box, random bf16 inputs.

    python tools/vendor_attn_ref.py [--reps 10]"""
import argparse
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from jepa_amd.hip import ops  # noqa: E402

SHAPES = [("target hd64", 24, 1568, 16, 64), ("context m0 hd64", 24, 366, 16, 64), ("predictor m0 hd24", 24, 1113, 16, 24),
          ("predictor m1 hd24", 24, 1208, 16, 24), ("ViT-H hd80", 8, 1568, 16, 80), ("ViT-H 384 hd80", 2, 4608, 16, 80)]


def timeit(fn, reps):
    for _ in range(2):
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
    ap.add_argument("--reps", type=int, default=10)
    args = ap.parse_args()
    g = torch.Generator(device="cuda").manual_seed(0)
    print(f"{'shape':20s} | ours fwd / bwd TF/s | vendor SDPA fwd / fwd+bwd-derived bwd TF/s")
    for tag, B, S, H, hd in SHAPES:
        qkv = torch.randn(B * S, 3 * H * hd, device="cuda", generator=g).to(torch.bfloat16)
        dout = torch.randn(B * S, H * hd, device="cuda", generator=g).to(torch.bfloat16)
        scale = hd ** -0.5
        o, lse = ops.attn_fwd(qkv, B, S, H, hd, scale)
        fl = 4.0 * B * H * S * S * hd
        tf = timeit(lambda: ops.attn_fwd(qkv, B, S, H, hd, scale), args.reps)
        tb = timeit(lambda: ops.attn_bwd(qkv, o, dout, lse, B, S, H, hd, scale), args.reps)
        # the reference's own call: q, k, v [B, H, S, hd] views of the packed projection (modules.py:63-69)
        q, k, v = qkv.view(B, S, 3, H, hd).permute(2, 0, 3, 1, 4)
        q, k, v = q.contiguous(), k.contiguous(), v.contiguous()
        try:
            vf = timeit(lambda: F.scaled_dot_product_attention(q, k, v), args.reps)
            qg, kg, vg = (t.clone().requires_grad_(True) for t in (q, k, v))
            do4 = dout.view(B, S, H, hd).permute(0, 2, 1, 3).contiguous()

            def fb():
                out = F.scaled_dot_product_attention(qg, kg, vg)
                out.backward(do4)
                qg.grad = kg.grad = vg.grad = None
            vfb = timeit(fb, args.reps)
            vb = vfb - vf
            vend = f"{fl / vf / 1e9:7.0f} / {2 * fl / vb / 1e9:7.0f}"
        except Exception as ex:   # noqa: BLE001
            vend = f"unavailable ({type(ex).__name__})"
        print(f"{tag:20s} | {fl / tf / 1e9:7.0f} / {2 * fl / tb / 1e9:7.0f}     | {vend}", flush=True)


if __name__ == "__main__":
    main()
