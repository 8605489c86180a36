#!/usr/bin/env python
"""Cost of the residual operand in the GEMM epilogue: the same GEMM with and without `residual` (and with bias only)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from jepa_amd.hip import ops  # noqa: E402


def timeit(fn, reps=20):
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
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(0)
    for tag, M, N, K in (("tgt proj", 37632, 1024, 1024), ("tgt fc2", 37632, 1024, 4096), ("ctx proj", 10560, 1024, 1024),
                         ("ctx fc2", 10560, 1024, 4096), ("prd fc2", 58560, 384, 1536)):
        a = torch.randn(M, K, device=dev, generator=g).to(torch.bfloat16)
        w = torch.randn(N, K, device=dev, generator=g).to(torch.bfloat16)
        b = torch.randn(N, device=dev, generator=g)
        r = torch.randn(M, N, device=dev, generator=g).to(torch.bfloat16)
        out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        t0 = timeit(lambda: ops.gemm_nt(a, w, bias=b, out=out))
        t1 = timeit(lambda: ops.gemm_nt(a, w, bias=b, residual=r, out=out))
        fl = 2.0 * M * N * K
        print(f"{tag:9s} {M}x{N}x{K}: bias only {t0:7.1f} us ({fl / t0 / 1e6:6.0f} TF/s) | + residual {t1:7.1f} us ({fl / t1 / 1e6:6.0f} TF/s)"
              f" | +{t1 - t0:5.1f} us", flush=True)


if __name__ == "__main__":
    main()
