#!/usr/bin/env python
"""Per-shape HBM traffic audit of the step's GEMMs: PMC bytes vs algorithmic bytes.

Run mode (on the GPU, under rocprofv3 --pmc FETCH_SIZE  and again under --pmc WRITE_SIZE):
    python tools/gemm_traffic.py run
  launches, for every GEMM shape of the ViT-L B=24 step, one marker kernel (vj_probe_copy) followed by ONE GEMM launch
  (default kernel selection), operands freshly written (so not L2/MALL resident from an earlier launch of the same data:
  the 256 MiB Infinity Cache still absorbs re-reads inside a launch, which is the point of the audit).
Summary mode:
    python tools/gemm_traffic.py summary <fetch_dir> <write_dir> <out.md>
  splits both counter CSVs at the markers and prints, per shape: algorithmic bytes (operands read once + outputs
  written once), FETCH_SIZE x2 (gfx950 correction, MI355X_MICROARCH.md HBM section) + WRITE_SIZE, and their ratio.
"""
import csv
import glob
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

SHAPES = [  # (tag, M, N, K, epilogue, extra operand): the step's shapes (profiles/r02_gemm_shapes_vitl16.md)
    ("tgt qkv", 37632, 3072, 1024, 0, None), ("tgt proj +res", 37632, 1024, 1024, 0, "res"),
    ("tgt fc1 gelu", 37632, 4096, 1024, 1, None), ("tgt fc2 +res", 37632, 1024, 4096, 0, "res"),
    ("tgt patch", 37632, 1024, 1536, 0, None),
    ("ctx qkv", 10560, 3072, 1024, 0, None), ("ctx proj +res", 10560, 1024, 1024, 0, "res"),
    ("ctx fc1 gelu+u", 10560, 4096, 1024, 1, "aux_out"), ("ctx fc2 +res", 10560, 1024, 4096, 0, "res"),
    ("ctx dfc2 dgelu", 10560, 4096, 1024, 2, "aux_in"), ("ctx dfc1", 10560, 1024, 4096, 0, None),
    ("ctx dqkv", 10560, 1024, 3072, 0, None),
    ("prd qkv", 58560, 1152, 384, 0, None), ("prd proj +res", 58560, 384, 384, 0, "res"),
    ("prd fc1 gelu+u", 58560, 1536, 384, 1, "aux_out"), ("prd fc2 +res", 58560, 384, 1536, 0, "res"),
    # weight gradients as the step launches them since round 3: the four of a block in ONE grouped launch (vj_gemm_bf16_tn_grouped);
    # M = sum N1*N2 / N of the group (the row count of the per-shape tables), N, K = tokens
    ("wg ctx block (4 in one launch)", 3072, 4096, 10560, 4, (1024, 4096)), ("wg prd block (4 in one launch)", 1152, 1536, 58560, 4, (384, 1536)),
]


def group_problems(D, Dh):
    """(N_out, K_in) of a block's four Linears: qkv, proj, fc1, fc2."""
    return [(3 * D, D), (D, D), (Dh, D), (D, Dh)]


def algorithmic_bytes(M, N, K, epi, extra):
    if epi == 4:   # grouped weight gradients: eight distinct token-major operands read once, four fp32 gradients written once
        return sum(2 * K * (n1 + n2) + 4 * n1 * n2 for n1, n2 in group_problems(*extra))
    b = 2 * M * K + 2 * N * K                      # operands, read once
    b += 4 * M * N if epi == 3 else 2 * M * N      # output, written once
    if extra in ("res", "aux_in"):
        b += 2 * M * N
    if extra == "aux_out":
        b += 2 * M * N
    if epi != 3:
        b += 4 * N                                  # bias
    return b


def run():
    import torch
    from jepa_amd.hip import ops
    from jepa_amd.hip.lib import check, load_library
    lib = load_library()
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(0)
    mark_src = torch.zeros(4096, dtype=torch.uint8, device=dev)
    mark_dst = torch.zeros(4096, dtype=torch.uint8, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    for tag, M, N, K, epi, extra in SHAPES:
        if epi == 4:
            probs = [(torch.randn(K, n1, device=dev, generator=g).to(torch.bfloat16), torch.randn(K, n2, device=dev, generator=g).to(torch.bfloat16),
                      torch.empty(n1, n2, device=dev, dtype=torch.float32)) for n1, n2 in group_problems(*extra)]
            flush = torch.randn(96 << 20, device=dev, generator=g)
            del flush
            torch.cuda.synchronize()
            check(lib.vj_probe_copy(mark_src.data_ptr(), mark_dst.data_ptr(), 4096, st), "marker")
            ops.gemm_wgrad_tn_grouped(probs, alpha=1.0, beta=0.0)
            torch.cuda.synchronize()
            continue
        A = torch.randn(M, K, device=dev, generator=g).to(torch.bfloat16)
        B = (torch.randn(N, K, device=dev, generator=g) * 0.05).to(torch.bfloat16)
        bias = torch.randn(N, device=dev, generator=g)
        other = torch.randn(M, N, device=dev, generator=g).to(torch.bfloat16) if extra else None
        out = torch.empty(M, N, device=dev, dtype=torch.float32 if epi == 3 else torch.bfloat16)
        if epi == 3:   # weight gradient, transpose-free route (the default): dY [T, N_out], X [T, K_in] token-major
            dY, X = A.t().contiguous(), B.t().contiguous()
        flush = torch.randn(96 << 20, device=dev, generator=g)   # 384 MB: pushes the operands out of the Infinity Cache
        del flush
        torch.cuda.synchronize()
        check(lib.vj_probe_copy(mark_src.data_ptr(), mark_dst.data_ptr(), 4096, st), "marker")
        if epi == 3:
            ops.gemm_wgrad_tn(dY, X, out)
        elif epi == 1:
            ops.gemm_nt(A, B, out=out, bias=bias, aux_out=other, epilogue=1)
        elif epi == 2:
            ops.gemm_nt(A, B, out=out, aux_in=other, epilogue=2)
        else:
            ops.gemm_nt(A, B, out=out, bias=bias, residual=other if extra == "res" else None)
        torch.cuda.synchronize()


def per_shape(d, counter):
    rows = []
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            rows += [r for r in csv.DictReader(fh) if r["Counter_Name"] == counter]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    out, cur = [], None
    for r in rows:
        name = r["Kernel_Name"]
        if "probe_copy" in name:
            cur = {"kb": 0.0, "kernels": [], "ns": 0.0}
            out.append(cur)
        elif cur is not None and ("gemm" in name or "splitk_reduce" in name or "reduce_partials" in name):
            cur["kb"] += float(r["Counter_Value"])
            cur["kernels"].append(name.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")[:44])
            cur["ns"] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    return out


def summary(fd, wd, out_md):
    fe, wr = per_shape(fd, "FETCH_SIZE"), per_shape(wd, "WRITE_SIZE")
    assert len(fe) == len(wr) == len(SHAPES), (len(fe), len(wr), len(SHAPES))
    lines = ["# Per-shape HBM traffic of the step's GEMMs, PMC vs algorithmic (default kernel selection: persistent 8-phase, TN weight gradients)", "",
             "`rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` (separate passes) over `python tools/gemm_traffic.py run`: one "
             "launch per shape, operands evicted from the Infinity Cache first.  Bytes = FETCH_SIZE x 2 (gfx950 counts 128-byte "
             "requests at 64 bytes) + WRITE_SIZE, both reported in KB.  Algorithmic = every operand read once + every output "
             "written once.  Ratio > 1 = re-reads that reached HBM; split-K weight gradients include their fp32 partials and the "
             "slice reduction (written + read once each by construction).", "",
             "| shape | M | N | K | kernels | algorithmic MB | FETCHx2 MB | WRITE MB | PMC / algorithmic | us (profiled) | "
             "achieved HBM GB/s |", "|---|---|---|---|---|---|---|---|---|---|---|"]
    for (tag, M, N, K, epi, extra), f, w in zip(SHAPES, fe, wr):
        alg = algorithmic_bytes(M, N, K, epi, extra)
        fb, wb = 2 * f["kb"] * 1024, w["kb"] * 1024
        ks = "+".join(sorted(set(f["kernels"])))
        lines.append(f"| {tag} | {M} | {N} | {K} | `{ks}` | {alg / 1e6:.1f} | {fb / 1e6:.1f} | {wb / 1e6:.1f} | "
                     f"{(fb + wb) / alg:.2f} | {f['ns'] / 1e3:.0f} | {(fb + wb) / f['ns']:.0f} |")
    with open(out_md, "w") as fh:
        fh.write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run()
    else:
        summary(*sys.argv[2:5])
