#!/usr/bin/env python
"""Per-shape table of the step's GEMM / attention launches from `bench.py --gemm-csv FILE` (one line per launch,
HIP-event durations on the launch stream, serial pass): shape, launches per step, mean us, TF/s, share of family time.

    python tools/gemm_table.py gpurun_out/gemm.csv 3 > profiles/r02_gemm_shapes.md      (3 = instrumented steps)
"""
import collections
import csv
import sys

FAM = {0: "GEMM", 1: "attention fwd", 2: "attention bwd"}
EPI = {0: "bf16 (+bias/+res)", 1: "bias+GELU", 2: "dGELU", 3: "wgrad fp32 split-K", 4: "4 wgrads of a block, one launch (rows = sum N1*N2 / N)"}


def main():
    path, n_steps = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 1
    acc = collections.OrderedDict()
    for r in csv.DictReader(open(path)):
        fam, tag = int(r["family"]), int(r["tag"])
        key = (fam, tag, int(r["m"]), int(r["n"]), int(r["k"]))
        a = acc.setdefault(key, [0, 0.0, 0.0])
        a[0] += 1
        a[1] += float(r["us"])
        a[2] += float(r["flop"]) if r.get("flop") else 0.0   # (round 4: the launch's own FLOP count; merged-segment attention launches)
    tot = collections.Counter()
    for (fam, *_), (n, us, _fl) in acc.items():
        tot[fam] += us
    for fam in sorted(FAM):
        rows = [(k, v) for k, v in acc.items() if k[0] == fam]
        if not rows:
            continue
        print(f"\n### {FAM[fam]}: {tot[fam] / n_steps / 1e3:.2f} ms per step\n")
        if fam == 0:
            print("| M | N | K | epilogue | launches/step | mean us | TF/s | % of GEMM time |")
            print("|---|---|---|---|---|---|---|---|")
        else:
            print("| B | S (longest segment of the launch) | H | head_dim | launches/step | mean us | TF/s | % of family time |")
            print("|---|---|---|---|---|---|---|---|")
        rows.sort(key=lambda kv: -kv[1][1])
        for (f, tag, m, n, k), (cnt, us, fl) in rows:
            if fam == 0:
                flop = 2.0 * m * n * k
                label = EPI.get(tag, str(tag))
            else:
                flop = (4.0 if fam == 1 else 8.0) * m * k * n * n * tag   # B*H*S*S*hd
                label = str(tag)
            tf = fl / us / 1e6 if fl > 0 else flop * cnt / us / 1e6
            print(f"| {m} | {n} | {k} | {label} | {cnt / n_steps:.1f} | {us / cnt:.1f} | "
                  f"{tf:.0f} | {100 * us / tot[fam]:.1f} |")


if __name__ == "__main__":
    main()
