#!/usr/bin/env python
"""LayerNorm forward / backward micro-benchmark on the step's row counts (HBM-bound: bytes moved / time)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from jepa_amd.hip import ops  # noqa: E402


def main():
    dev = "cuda"
    for rows, D in ((37632, 1024), (11392, 1024), (27848, 384)):
        x = torch.randn(rows, D, device=dev).to(torch.bfloat16)
        g = torch.randn(D, device=dev)
        b = torch.randn(D, device=dev)
        for _ in range(3):
            ops.layernorm_fwd(x, g, b, 1e-6, save_stats=True)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(50):
            ops.layernorm_fwd(x, g, b, 1e-6, save_stats=True)
        e.record()
        torch.cuda.synchronize()
        us = s.elapsed_time(e) * 1e3 / 50
        print(f"LN fwd rows={rows} D={D}: {us:6.1f} us  {rows * D * 4 / us / 1e6:5.2f} TB/s")


if __name__ == "__main__":
    main()
