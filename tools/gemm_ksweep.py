#!/usr/bin/env python
"""Fixed overhead vs K-loop time of the 8-phase GEMM: time(K) at fixed M, N -> per-tile-round intercept and slope."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from jepa_amd.hip import ops  # noqa: E402


def main():
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(0)
    from jepa_amd.hip.lib import set_option
    import itertools
    shapes = ((37632, 3072), (37632, 1024), (55680, 1536))
    combos = list(itertools.product(shapes, (0, 1), (0, 1)))
    if os.environ.get("VJ_LIB_VARIANT"):   # A/B builds (e.g. skipreads): only the one-tile kernel on the first shape
        combos = [(shapes[0], 0, 0), (shapes[0], 0, 1)]
        print(f"=== library variant {os.environ['VJ_LIB_VARIANT']}")
    for (M, N), persist, dbg in combos:
        set_option("gemm_dbg", dbg)
        set_option("gemm_persist", persist)
        print(f"--- M={M} N={N} " + ("persistent kernel (gemm8p.hip)" if persist else "one tile per workgroup (gemm8.hip)") +
              (", WITHOUT the epilogue (gemm_dbg=1)" if dbg & 1 else ""))
        rounds = -(-((M + 255) // 256 * ((N + 255) // 256)) // 256)
        pts = []
        for K in (256, 384, 512, 1024, 2048, 4096):
            A = torch.randn(M, K, device=dev, generator=g).to(torch.bfloat16)
            B = (torch.randn(N, K, device=dev, generator=g) * 0.05).to(torch.bfloat16)
            bias = torch.randn(N, device=dev, generator=g)
            out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
            for _ in range(3):
                ops.gemm_nt(A, B, out=out, bias=bias, flags=(2 << 4) | (3 << 6))
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(20):
                ops.gemm_nt(A, B, out=out, bias=bias, flags=(2 << 4) | (3 << 6))
            e.record()
            torch.cuda.synchronize()
            us = s.elapsed_time(e) * 1e3 / 20
            pts.append((K, us))
            print(f"M={M} N={N} K={K:5d}: {us:8.1f} us  {2 * M * N * K / us / 1e6:7.1f} TF/s  per round {us / rounds:6.2f} us "
                  f"({rounds} rounds)")
        (k0, t0), (k1, t1) = pts[-3], pts[-1]
        slope = (t1 - t0) / (k1 - k0)
        print(f"  -> slope {slope * 64 / rounds:.3f} us per K-tile(64) per round, intercept {(t0 - slope * k0) / rounds:.2f} us "
              f"per round")


if __name__ == "__main__":
    main()
