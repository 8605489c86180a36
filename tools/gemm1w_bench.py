#!/usr/bin/env python
"""K-sweep of the EXPERIMENTAL one-wave-per-SIMD GEMM (csrc/gemm1w.hip) against the production kernels on the same operands:
time(K) at fixed M, N -> K-loop slope (us per 64 of K per tile round) and per-round intercept, with and without output traffic."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from jepa_amd.hip import ops  # noqa: E402
from jepa_amd.hip.lib import set_option  # noqa: E402


def timed(fn, reps=20):
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
    g = torch.Generator(device=dev).manual_seed(0)
    for M, N in ((37632, 3072), (37632, 1024), (10560, 1024), (8192, 8192)):
        tiles = ((M + 255) // 256) * ((N + 255) // 256)
        rounds = -(-tiles // 256)
        for name in ("1w", "1w no-store", "persistent", "persistent no-store", "one-tile"):
            pts = []
            for K in (256, 512, 1024, 2048, 4096):
                A = torch.randn(M, K, device=dev, generator=g).to(torch.bfloat16)
                B = (torch.randn(N, K, device=dev, generator=g) * 0.05).to(torch.bfloat16)
                bias = torch.randn(N, device=dev, generator=g)
                out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
                set_option("gemm_dbg", 1 if name == "persistent no-store" else 0)
                set_option("gemm_persist", 0 if name == "one-tile" else 1)
                if name.startswith("1w"):
                    fn = lambda: ops.gemm_nt_1w(A, B, bias=bias, out=out, dbg=1 if "no-store" in name else 0)   # noqa: E731
                else:
                    fn = lambda: ops.gemm_nt(A, B, out=out, bias=bias, flags=(2 << 4) | (3 << 6))   # noqa: E731
                us = timed(fn)
                pts.append((K, us))
            set_option("gemm_dbg", 0)
            set_option("gemm_persist", 1)
            (k0, t0), (k1, t1) = pts[-3], pts[-1]
            slope = (t1 - t0) / (k1 - k0)
            print(f"M={M} N={N} {name:20s} " + " ".join(f"K={k}: {2 * M * N * k / u / 1e6:6.0f} TF/s" for k, u in pts) +
                  f" | slope {slope * 64 / rounds:.3f} us per 64-K-tile per round, intercept {(t0 - slope * k0) / rounds:.2f} us "
                  f"({rounds} rounds of 256 tiles)", flush=True)


if __name__ == "__main__":
    main()
