#!/usr/bin/env python
"""HBM traffic per kernel from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; `--output-format csv`).

    python tools/pmc_summary.py <fetch_dir> <write_dir> <out.md> <out.json> [title]

Units / corrections (MI355X_MICROARCH.md, HBM section): both counters are in KB; on gfx950 FETCH_SIZE counts 128-byte
requests at 64 bytes, so reads are doubled; WRITE_SIZE is taken as reported."""
import collections
import csv
import glob
import json
import os
import re
import sys


def short(name):
    name = re.sub(r"^void ", "", name).replace("(anonymous namespace)::", "")
    name = name.split("(")[0]
    return name[:90]


def collect(d, counter):
    acc = collections.defaultdict(lambda: [0, 0.0])
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if row["Counter_Name"] != counter:
                    continue
                a = acc[short(row["Kernel_Name"])]
                a[0] += 1
                a[1] += float(row["Counter_Value"])
    return acc


def main():
    fd, wd, out_md, out_json = sys.argv[1:5]
    title = sys.argv[5] if len(sys.argv) > 5 else "HBM traffic per kernel (PMC)"
    fetch, write = collect(fd, "FETCH_SIZE"), collect(wd, "WRITE_SIZE")
    rows = []
    for k, (n, f) in fetch.items():
        wn, w = write.get(k, (0, 0.0))
        if n == 0:
            continue
        fk, wk = f / n, (w / wn if wn else 0.0)
        rows.append((k, n, fk, wk, (2 * fk + wk) * 1024))
    rows.sort(key=lambda r: -r[1] * r[4])
    with open(out_md, "w") as fh:
        fh.write(f"# {title}\n\n")
        fh.write("Separate `rocprofv3 --kernel-trace --pmc FETCH_SIZE|WRITE_SIZE --output-format csv` passes of "
                 "`python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline-pass`.  KB per launch; reads x2 "
                 "(gfx950 FETCH_SIZE counts 128-byte requests at 64 bytes), writes as reported.\n\n")
        fh.write("| kernel | launches | FETCH_SIZE KB/launch (raw) | x2 | WRITE_SIZE KB/launch | HBM bytes/launch |\n")
        fh.write("|---|---|---|---|---|---|\n")
        for k, n, fk, wk, b in rows[:28]:
            fh.write(f"| `{k}` | {n} | {fk:.0f} | {2 * fk:.0f} | {wk:.0f} | {b:.3e} |\n")
    with open(out_json, "w") as fh:
        json.dump({k: dict(launches=n, fetch_kb_raw=fk, write_kb=wk, hbm_bytes_per_launch=b) for k, n, fk, wk, b in rows},
                  fh, indent=1)


if __name__ == "__main__":
    main()
