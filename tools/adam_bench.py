#!/usr/bin/env python
"""AdamW + EMA + bf16 re-cast kernel on ViT-L-sized arenas (304 M encoder parameters with a target, 22 M predictor parameters without):
bytes moved / time (the unroll-by-two and non-temporal variants of round 4 measured the same and were removed:
profiles/r04_adam_variants.md)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from jepa_amd.hip import ops  # noqa: E402


def main():
    dev = "cuda"
    for n, with_tgt in ((303_000_000 // 1024 * 1024, True), (22_000_000 // 1024 * 1024, False)):
        p, g, m, v = (torch.randn(n, device=dev) * 0.02 for _ in range(4))
        v.abs_()
        t = p.clone() if with_tgt else None
        pb = torch.empty(n, dtype=torch.bfloat16, device=dev)
        tb = torch.empty(n, dtype=torch.bfloat16, device=dev) if with_tgt else None
        bytes_moved = n * (4 * 4 + 3 * 4 + 2 + ((4 + 4 + 2) if with_tgt else 0))
        for var in (0, 0):
            for _ in range(2):
                ops.adamw_ema(p, g, m, v, pb, t, tb, 1e-4, 0.05, 0.9, 0.999, 1e-8, 5, 1.0, 0.998)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(10):
                ops.adamw_ema(p, g, m, v, pb, t, tb, 1e-4, 0.05, 0.9, 0.999, 1e-8, 5, 1.0, 0.998)
            e.record()
            torch.cuda.synchronize()
            us = s.elapsed_time(e) * 1e3 / 10
            print(f"adamw n={n} target={with_tgt}: {us:8.1f} us  {bytes_moved / us / 1e6:5.2f} TB/s", flush=True)


if __name__ == "__main__":
    main()
