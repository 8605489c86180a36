#!/usr/bin/env python
"""LayerNorm forward / backward micro-benchmark on the step's row counts (HBM-bound: bytes moved / time).
Backward = the chains' variant (dx + residual gradient in, column partials of dgamma | dbeta | dx, one reduction launch)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from jepa_amd.hip import ops  # noqa: E402


def timed(fn, reps=50):
    for _ in range(3):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / reps


def main():
    dev = "cuda"
    for rows, D in ((37632, 1024), (10560, 1024), (9024, 1024), (58560, 384), (52800, 384), (10560, 1280)):
        x = torch.randn(rows, D, device=dev).to(torch.bfloat16)
        g = torch.randn(D, device=dev)
        b = torch.randn(D, device=dev)
        us = timed(lambda: ops.layernorm_fwd(x, g, b, 1e-6, save_stats=True))
        print(f"LN fwd rows={rows} D={D}: {us:6.1f} us  {rows * D * 4 / us / 1e6:5.2f} TB/s")
        _, mean, rstd = ops.layernorm_fwd(x, g, b, 1e-6, save_stats=True)
        dy = torch.randn(rows, D, device=dev).to(torch.bfloat16)
        dres = torch.randn(rows, D, device=dev).to(torch.bfloat16)
        dg, db, ds = (torch.zeros(D, device=dev) for _ in range(3))
        us = timed(lambda: ops.layernorm_bwd(dy, x, g, mean, rstd, dg, db, dres=dres, dxsum=ds))
        print(f"LN bwd rows={rows} D={D}: {us:6.1f} us  {rows * D * 8 / us / 1e6:5.2f} TB/s (x, dy, dres in, dx out; + reduction launch)")

if __name__ == "__main__":
    main()
