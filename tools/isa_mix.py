#!/usr/bin/env python
"""Static look at what hipcc emitted for a kernel (no GPU needed): resources from the metadata and, per basic block,
the instruction mix (MFMA / VALU / transcendental / packed / LDS / VMEM / SALU / waits).

    python tools/isa_mix.py jepa_amd/csrc/attention.hip attn_bwd_dq_kernel<32> [--min 40]

The file is compiled device-only to assembly with the flags jepa_amd/build.py uses.  Blocks with fewer than --min
instructions are skipped; loop bodies are the blocks whose last branch goes backwards.  Every block also gets a pipe-time estimate from
the instruction costs measured on the hardware (COST below): what the SIMD needs for the block at full occupancy is matrix + vector
cycles (the two pipes add on gfx950), the LDS pipe runs beside them with 16 cycles of a SIMD's share per read / write instruction."""
import argparse
import os
import re
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from jepa_amd import build as vb  # noqa: E402

TRANS = ("v_exp_", "v_log_", "v_rcp_", "v_rsq_", "v_sqrt_", "v_sin_", "v_cos_")

# Cycles a wave64 instruction costs its SIMD at full occupancy, measured with tools/probes/valu_probe.hip on an MI355X
# (profiles/r03_valu_mfma_probe.md, r03_attn_dkdv_kt2.md).  The matrix and the vector pipe of a SIMD ADD; the LDS pipe is shared by the
# four SIMDs of a CU (4 cycles per wave-instruction CU-wide = 16 of a SIMD's share) and runs beside them.
COST = {"mfma": 16.0, "valu": 2.65, "vpk": 4.35, "trans": 8.2, "acc_mov": 2.65, "lds": 16.0}
COST_OP = {"v_cvt_pk_bf16_f32": 4.4, "v_max3_f32": 4.3, "v_permlane32_swap_b32": 8.1, "v_permlane16_swap_b32": 8.1}


def pipe_cycles(ins):
    """(matrix, vector, LDS-share) cycles of a list of instructions for one wave: the SIMD needs matrix + vector, the LDS pipe the third."""
    m = v = l = 0.0
    for i in ins:
        op = i.split()[0]
        c = classify(op)
        if c == "mfma":
            m += COST["mfma"] * (2.0 if "32x32" in op else 1.0)
        elif c in ("valu", "vpk", "trans", "acc_mov"):
            v += COST_OP.get(op.replace("_e32", "").replace("_e64", ""), COST[c])
        elif c == "lds":
            l += COST["lds"]
    return m, v, l


def classify(op):
    if op.startswith("v_mfma") or op.startswith("v_smfmac"):
        return "mfma"
    if op.startswith(TRANS):
        return "trans"
    if op.startswith("v_pk_"):
        return "vpk"
    if op.startswith("v_accvgpr"):
        return "acc_mov"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith("s_waitcnt"):
        return "wait"
    if op.startswith("s_barrier"):
        return "barrier"
    if op.startswith(("s_cbranch", "s_branch")):
        return "branch"
    if op.startswith("s_"):
        return "salu"
    return "other"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("src")
    ap.add_argument("kernel", help="substring of the demangled kernel name")
    ap.add_argument("--min", type=int, default=40)
    ap.add_argument("--dump", action="store_true", help="print the instructions of the listed blocks")
    a = ap.parse_args()
    base = os.path.basename(a.src)
    out = f"/tmp/isa_{os.path.splitext(base)[0]}.s"
    cmd = [vb._hipcc()] + vb.CXXFLAGS + vb.EXTRA_FLAGS.get(base, []) + ["-x", "hip", "--cuda-device-only", "-S", a.src, "-o", out]
    subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
    text = open(out).read()
    # symbol -> demangled
    syms = re.findall(r"^\s+\.globl\s+(\S+)", text, re.M)
    dem = subprocess.run(["c++filt"] + syms, capture_output=True, text=True).stdout.splitlines()
    pick = [s for s, d in zip(syms, dem) if a.kernel in d and "(" in d]
    if not pick:
        print("no kernel matches; candidates:\n  " + "\n  ".join(sorted(set(d.split("(")[0] for d in dem))))
        return 1
    for sym in pick:
        d = dem[syms.index(sym)].split("(")[0]
        meta = re.search(r"\.name:\s+" + re.escape(sym) + r"\n(.*?)\.symbol:", text, re.S)
        md = dict(re.findall(r"\.(\w+):\s+(\d+)", meta.group(1))) if meta else {}
        body = re.search(r"^" + re.escape(sym) + r":[^\n]*\n(.*?)^\.Lfunc_end", text, re.S | re.M).group(1)
        print(f"== {d}: vgpr {md.get('vgpr_count')} agpr {md.get('agpr_count')} sgpr {md.get('sgpr_count')} "
              f"spill {md.get('vgpr_spill_count')} lds {md.get('group_segment_fixed_size')}")
        blocks, cur, name, order = {}, [], "entry", []
        for ln in body.splitlines():
            s = ln.strip()
            m = re.match(r"^(\.LBB\S+):", s)
            if m:
                blocks[name] = cur
                order.append(name)
                name, cur = m.group(1), []
                continue
            if not s or s.startswith((";", ".", "//")):
                continue
            cur.append(s.split(";")[0].strip())
        blocks[name] = cur
        order.append(name)
        idx = {n: i for i, n in enumerate(order)}
        for n in order:
            ins = blocks[n]
            if len(ins) < a.min:
                continue
            mix = {}
            for i in ins:
                c = classify(i.split()[0])
                mix[c] = mix.get(c, 0) + 1
            back = any(i.startswith("s_cbranch") and idx.get(i.split()[-1], 1 << 30) <= idx[n] for i in ins)
            tot_v = mix.get("valu", 0) + mix.get("vpk", 0) + mix.get("trans", 0) + mix.get("acc_mov", 0)
            pm, pv, pl = pipe_cycles(ins)
            print(f"  {n:12s} {'LOOP' if back else '    '} n={len(ins):5d}  " + "  ".join(f"{k}={v}" for k, v in sorted(mix.items()))
                  + f"  | VALU-all/MFMA = {tot_v / max(1, mix.get('mfma', 0)):.2f}"
                  + f"  | pipe cycles: matrix {pm:.0f} + vector {pv:.0f} = {pm + pv:.0f}, LDS share {pl:.0f}")
            if a.dump:
                for i in ins:
                    print("      " + i)
    return 0


if __name__ == "__main__":
    sys.exit(main())
