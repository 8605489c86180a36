#!/usr/bin/env python
"""LDS read throughput per CU on gfx950 for the three read forms the GEMM / attention kernels use (HIP events around a
one-workgroup-per-CU launch; bytes / s / CU and bytes per shader clock at the sclk read from amdsmi if available)."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from jepa_amd.hip.lib import check, load_library  # noqa: E402


def main():
    lib = load_library()
    n_wgs, iters = 256, 20000
    out = torch.zeros(n_wgs * 8, dtype=torch.int64, device="cuda")
    names = {0: ("ds_read_b128", 16), 2: ("ds_read_b64", 8), 1: ("ds_read_b64_tr_b16 (TN fragment addressing)", 8)}
    for mode in (0, 2, 1):
        for wgs in (1, n_wgs):
            for _ in range(2):
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                check(lib.vj_probe_lds_bw(ctypes.c_void_p(out.data_ptr()), mode, iters, wgs, None), "lds")
                e.record()
                torch.cuda.synchronize()
            ms = s.elapsed_time(e)
            nbytes = 8 * iters * 8 * 64 * names[mode][1]
            instr = 8 * iters * 8
            print(f"{names[mode][0]:45s} {wgs:3d} workgroup(s): {ms * 1e3:8.1f} us, {nbytes / ms / 1e6:7.1f} GB/s per CU, "
                  f"{ms * 1e6 / instr * 2.4:5.2f} cycles @2.4 GHz per wave-instruction (8 waves share the LDS)", flush=True)


if __name__ == "__main__":
    main()
