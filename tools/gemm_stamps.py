#!/usr/bin/env python
"""Where a tile of the persistent GEMM spends its time: option gemm_dbg = 4 runs the phase-stamping copy of the default kernel
(gemm8p.hip, STAMP): waves 0 and 4 of every workgroup read s_memtime at the phase boundaries of each tile and leave the sums in the
first bytes of C.  Printed per shape and epilogue: the mean per tile of

    start   first barrier of the tile -> first MFMA section
    kloop   the K loop
    dma     s_waitcnt vmcnt(0) for the next tile's prefetched parts (issued during the last K-tile)
    epi     bias (+ row operand) latency, convert, stage, store issue
    skew    epilogue end -> past the next tile's first barrier (what the slowest of the eight waves adds)

in microseconds (the counter's rate is calibrated on the launch's event time) for the early and the late wave group."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from jepa_amd.hip import ops  # noqa: E402
from jepa_amd.hip.lib import set_option  # noqa: E402

NAMES = ("start", "kloop", "dma", "epi", "skew")


def run(tag, M, N, K, kind, dev="cuda"):
    g = torch.Generator(device=dev).manual_seed(0)
    A = torch.randn(M, K, device=dev, generator=g).to(torch.bfloat16)
    W = (torch.randn(N, K, device=dev, generator=g) * 0.05).to(torch.bfloat16)
    bias = torch.randn(N, device=dev, generator=g)
    res = torch.randn(M, N, device=dev, generator=g).to(torch.bfloat16)
    aux = torch.rand(M, N, device=dev, generator=g).to(torch.bfloat16)
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    auxo = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    fn = {
        "plain": lambda: ops.gemm_nt(A, W, out=out),
        "bias": lambda: ops.gemm_nt(A, W, bias=bias, out=out),
        "bias+res": lambda: ops.gemm_nt(A, W, bias=bias, residual=res, out=out),
        "gelu": lambda: ops.gemm_nt(A, W, bias=bias, out=out, epilogue=ops.EPI_GELU),
        "gelu+dgelu": lambda: ops.gemm_nt(A, W, bias=bias, out=out, aux_out=auxo, epilogue=ops.EPI_GELU),
        "dgelu": lambda: ops.gemm_nt(A, W, aux_in=aux, out=out, epilogue=ops.EPI_DGELU),
    }[kind]
    for _ in range(3):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10):
        fn()
    e.record()
    torch.cuda.synchronize()
    us_ref = s.elapsed_time(e) * 100.0
    old = set_option("gemm_dbg", 4)
    fn()
    torch.cuda.synchronize()
    s.record()
    fn()
    e.record()
    torch.cuda.synchronize()
    set_option("gemm_dbg", old)
    us = s.elapsed_time(e) * 1e3
    raw = out.view(-1).view(torch.int64)[: 256 * 2 * 8].reshape(256, 2, 8).cpu()
    ok = raw[:, :, 7] == 0x5354414D50
    nwg = int(ok[:, 0].sum())
    if nwg == 0:
        print(f"{tag:9s} {M}x{N}x{K} {kind:10s}: no stamps (the launch did not take the persistent 4-section kernel)")
        return
    r = raw[ok[:, 0]].double()
    tiles = r[:, :, 5]
    tot = r[:, :, :5].sum(-1)                      # cycles covered by the laps of a (workgroup, group)
    rate = float(tot.max()) / us                   # counter ticks per microsecond (the laps cover the launch but for its prologue)
    line = f"{tag:9s} {M}x{N}x{K} {kind:10s}: {us_ref:7.1f} us ({2.0 * M * N * K / us_ref / 1e6:5.0f} TF/s; stamped run {us:7.1f} us), {nwg} workgroups x {tiles[:, 0].mean():.2f} tiles, {rate:6.1f} ticks/us |"
    for grp in (0, 1):
        per = r[:, grp, :5].sum(0) / tiles[:, grp].sum() / rate
        line += " " + ("early" if grp == 0 else "late") + " " + " ".join(f"{n} {v:5.2f}" for n, v in zip(NAMES, per.tolist())) + " |"
    print(line, flush=True)


def main():
    cases = [
        ("tgt proj", 37632, 1024, 1024, ("plain", "bias", "bias+res")),
        ("tgt qkv", 37632, 3072, 1024, ("bias",)),
        ("tgt fc1", 37632, 4096, 1024, ("gelu",)),
        ("tgt fc2", 37632, 1024, 4096, ("bias+res",)),
        ("ctx fc1", 10496, 4096, 1024, ("gelu+dgelu",)),
        ("ctx dfc2", 10496, 4096, 1024, ("dgelu",)),
        ("prd qkv", 58368, 1152, 384, ("bias",)),
        ("prd proj", 58368, 384, 384, ("plain", "bias", "bias+res")),
        ("prd fc1", 58368, 1536, 384, ("gelu+dgelu",)),
        ("prd fc2", 58368, 384, 1536, ("bias+res",)),
        ("prd dfc2", 58368, 1536, 384, ("dgelu",)),
    ]
    # arguments: comma-separated gemm_epi_pre values, then any number of option settings "name=value" (e.g. gemm_raster=0 gemm_persist=2),
    # then optionally "only=<tag substring>"
    pres = [int(x) for x in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["4"])]
    only = None
    for kv in sys.argv[2:]:
        k, v = kv.split("=")
        if k == "only":
            only = v
        else:
            set_option(k, int(v))
    for pre in pres:
        set_option("gemm_epi_pre", pre)
        print(f"=== gemm_epi_pre = {pre} " + " ".join(sys.argv[2:]))
        for tag, M, N, K, kinds in cases:
            if only is not None and only not in tag:
                continue
            for kind in kinds:
                run(tag, M, N, K, kind)


if __name__ == "__main__":
    main()
