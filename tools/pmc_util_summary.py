#!/usr/bin/env python
"""MFMA / VALU / LDS utilisation per kernel from rocprofv3 SQ counter passes (`--pmc ... --output-format csv`).

    python tools/pmc_util_summary.py <out.md> <title> <pass_dir> [<pass_dir> ...]

Every pass directory holds one *_counter_collection.csv; counters of different passes are joined per kernel name
(the passes run the same command, so dispatch k of pass A is dispatch k of pass B).  Definitions used:
  mfma_util   = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024)   busy cycles of the 1024 matrix pipes over the
                dispatch's shader cycles (GRBM_GUI_ACTIVE is summed over the 8 XCDs; the MFMA counter counts cycles, 16
                per 16x16x32 bf16 MFMA: BUSY == 16 * SQ_INSTS_MFMA exactly in every row) = achieved / peak AT THE CLOCK THE
                KERNEL RAN AT (profiled passes clock ~2.0 GHz, MI355X_MICROARCH.md DVFS note)
  mfma_ops    = SQ_INSTS_VALU_MFMA_MOPS_BF16 (x512 flop) -> TFLOP/s over the dispatch's wall time
  valu_busy   = SQ_ACTIVE_INST_VALU * 4 / SQ_WAVE_CYCLES ... reported raw, both in quad-cycles
  lds_conf    = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE
"""
import collections
import csv
import glob
import os
import re
import sys


def short(name):
    name = re.sub(r"^void ", "", name).replace("(anonymous namespace)::", "").split("(")[0]
    return name[:70]


def main():
    out_md, title, dirs = sys.argv[1], sys.argv[2], sys.argv[3:]
    acc = collections.OrderedDict()
    for d in dirs:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            seen = set()
            with open(f) as fh:
                for row in csv.DictReader(fh):
                    k = short(row["Kernel_Name"])
                    if k.startswith("at::") or "elementwise" in k or "distribution" in k:
                        continue
                    a = acc.setdefault(k, collections.defaultdict(float))
                    a[row["Counter_Name"]] += float(row["Counter_Value"])
                    did = (d, row["Dispatch_Id"])
                    if did not in seen:
                        seen.add(did)
                        a["_ns@" + d] += float(row["End_Timestamp"]) - float(row["Start_Timestamp"])
                        a["_n@" + d] += 1
    lines = [f"# {title}", "",
             "| kernel | dispatches | mean us | MFMA util % | MFMA TF/s | wave-cyc: active VALU % | wait-any % | "
             "wait-inst % | LDS conflict % | VALU inst / MFMA inst | trans / VALU % |",
             "|---|---|---|---|---|---|---|---|---|---|---|"]
    for k, a in acc.items():
        ns = [v for kk, v in a.items() if kk.startswith("_ns@")]
        n = [v for kk, v in a.items() if kk.startswith("_n@")]
        if not ns:
            continue
        us = ns[0] / n[0] / 1e3
        gui = a.get("GRBM_GUI_ACTIVE", 0.0)
        # GRBM_GUI_ACTIVE is summed over the passes that collected it: normalise to one pass
        n_gui = sum(1 for d in dirs if any(True for _ in [0]) and ("_ns@" + d) in a)
        gui_pass = gui / max(1, n_gui)
        util = 100.0 * a.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (gui_pass / 8 * 1024) if gui_pass else float("nan")
        tf = a.get("SQ_INSTS_VALU_MFMA_MOPS_BF16", 0.0) * 512 / (ns[0] * 1e-9) / 1e12 if ns[0] else float("nan")
        wc = a.get("SQ_WAVE_CYCLES", 0.0)
        pct = lambda c: (100.0 * a.get(c, 0.0) / wc) if wc else float("nan")   # noqa: E731
        conf = 100.0 * a.get("SQ_LDS_BANK_CONFLICT", 0.0) / a["SQ_LDS_IDX_ACTIVE"] if a.get("SQ_LDS_IDX_ACTIVE") else 0.0
        vm = a.get("SQ_INSTS_VALU", 0.0) / a["SQ_INSTS_MFMA"] if a.get("SQ_INSTS_MFMA") else float("nan")
        tr = 100.0 * a.get("SQ_INSTS_VALU_TRANS_F32", 0.0) / a["SQ_INSTS_VALU"] if a.get("SQ_INSTS_VALU") else float("nan")
        lines.append(f"| `{k}` | {int(n[0])} | {us:.1f} | {util:.1f} | {tf:.0f} | {pct('SQ_ACTIVE_INST_VALU'):.1f} | "
                     f"{pct('SQ_WAIT_ANY'):.1f} | {pct('SQ_WAIT_INST_ANY'):.1f} | {conf:.1f} | {vm:.2f} | {tr:.1f} |")
    lines += ["", "Raw counter sums per kernel:", ""]
    for k, a in acc.items():
        raw = ", ".join(f"{c}={v:.4g}" for c, v in sorted(a.items()) if not c.startswith("_"))
        lines.append(f"* `{k}`: {raw}")
    with open(out_md, "w") as fh:
        fh.write("\n".join(lines) + "\n")
    print("\n".join(lines[:40]))


if __name__ == "__main__":
    main()
