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
    """For every value of option gemm_epi_pre (how the persistent kernel's epilogue requests its row operand): bias only | + residual, and
    the dGELU epilogue (row operand = the saved gelu') with the fused column sums."""
    from jepa_amd.hip.lib import set_option
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(0)
    pres = [int(x) for x in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["0", "1", "2"])]
    for tag, M, N, K in (("tgt proj", 37632, 1024, 1024), ("tgt fc2", 37632, 1024, 4096), ("ctx proj", 10560, 1024, 1024),
                         ("ctx fc2", 10560, 1024, 4096), ("ctx dqkv", 10560, 1024, 3072), ("prd proj", 58560, 384, 384),
                         ("prd fc2", 58560, 384, 1536), ("prd dqkv", 58560, 384, 1152)):
        a = torch.randn(M, K, device=dev, generator=g).to(torch.bfloat16)
        w = torch.randn(N, K, device=dev, generator=g).to(torch.bfloat16)
        b = torch.randn(N, device=dev, generator=g)
        r = torch.randn(M, N, device=dev, generator=g).to(torch.bfloat16)
        out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        fl = 2.0 * M * N * K
        t0 = timeit(lambda: ops.gemm_nt(a, w, bias=b, out=out))
        line = f"{tag:9s} {M}x{N}x{K}: bias only {t0:7.1f} us ({fl / t0 / 1e6:6.0f} TF/s) | + residual"
        for pre in pres:
            old = set_option("gemm_epi_pre", pre)
            t1 = timeit(lambda: ops.gemm_nt(a, w, bias=b, residual=r, out=out))
            set_option("gemm_epi_pre", old)
            line += f"  pre={pre} {t1:7.1f} us ({fl / t1 / 1e6:6.0f} TF/s, +{t1 - t0:5.1f})"
        print(line, flush=True)
    for tag, M, N, K in (("ctx dfc2", 10560, 4096, 1024), ("prd dfc2", 58560, 1536, 384)):
        a = torch.randn(M, K, device=dev, generator=g).to(torch.bfloat16)
        w = torch.randn(N, K, device=dev, generator=g).to(torch.bfloat16)
        aux = torch.rand(M, N, device=dev, generator=g).to(torch.bfloat16)
        out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        fl = 2.0 * M * N * K
        t0 = timeit(lambda: ops.gemm_nt(a, w, out=out))
        line = f"{tag:9s} {M}x{N}x{K}: plain     {t0:7.1f} us ({fl / t0 / 1e6:6.0f} TF/s) | dGELU+colsum"
        for pre in pres:
            old = set_option("gemm_epi_pre", pre)
            t1 = timeit(lambda: ops.gemm_dgelu_colsum(a, w, aux))
            set_option("gemm_epi_pre", old)
            line += f"  pre={pre} {t1:7.1f} us ({fl / t1 / 1e6:6.0f} TF/s, +{t1 - t0:5.1f})"
        print(line, flush=True)


if __name__ == "__main__":
    main()
