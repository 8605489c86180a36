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
    ap.add_argument("--sm", default="", help="comma list of attn_softmax option values to run one after the other (e.g. 0,1)")
    ap.add_argument("--opts", default="", help="semicolon list of option settings to run one after the other, e.g. ';attn_softmax=1'")
    ap.add_argument("--errors", action="store_true", help="also print rel-L2 of o / dq / dk / dv against fp32 SDPA (small B)")
    args = ap.parse_args()
    dev = "cuda"
    from jepa_amd.hip.lib import get_option, set_option
    sms = [int(v) for v in args.sm.split(",") if v.strip()] or [get_option("attn_softmax")]
    settings = args.opts.split(";") if args.opts else [""]
    for sm in sms:
        set_option("attn_softmax", sm)
        for st in settings:
            kv = [x.partition("=") for x in st.split(",") if x.strip()]
            olds = [(k.strip(), set_option(k.strip(), int(v))) for k, _, v in kv]
            print(f"--- attn_softmax = {sm} {st}", flush=True)
            run_shapes(args, dev)
            for k, old in olds:
                set_option(k, old)


def errors(ops, B, S, H, hd):
    """rel-L2 against fp32 SDPA on min(B, 2) samples"""
    import torch.nn.functional as F
    Bs = min(B, 2)
    g = torch.Generator(device="cuda").manual_seed(1)
    qkv = torch.randn(Bs * S, 3 * H * hd, device="cuda", generator=g).to(torch.bfloat16)
    dout = torch.randn(Bs * S, H * hd, device="cuda", generator=g).to(torch.bfloat16)
    o, lse = ops.attn_fwd(qkv, Bs, S, H, hd, hd ** -0.5)
    dqkv = ops.attn_bwd(qkv, o, dout, lse, Bs, S, H, hd, hd ** -0.5)
    x = qkv.float().requires_grad_(True)
    q, k, v = x.view(Bs, S, 3, H, hd).permute(2, 0, 3, 1, 4)
    ref = F.scaled_dot_product_attention(q, k, v).transpose(1, 2).reshape(Bs * S, H * hd)
    ref.backward(dout.float())
    rl = lambda a, b: float((a.float() - b).norm() / b.norm())
    gr, go = x.grad.view(Bs, S, 3, H, hd), dqkv.float().view(Bs, S, 3, H, hd)
    return [rl(o, ref)] + [rl(go[:, :, i], gr[:, :, i]) for i in range(3)]


def run_shapes(args, dev):
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
        if args.errors:
            e = errors(ops, B, S, H, hd)
            line += " | rel-L2 o %.2e dq %.2e dk %.2e dv %.2e" % tuple(e)
        print(line, flush=True)


if __name__ == "__main__":
    main()
