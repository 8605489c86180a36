#!/usr/bin/env python
"""Joules per FLOP of MFMA 16x16x32 vs 32x32x16 (bf16) at the package cap: host side of tools/probes/mfma_power_probe.hip.

Every case fills the chip (one workgroup per CU) with the same 64 x 64 x 32 block-step per wave and loops for `--seconds` while
package power and shader clock are sampled (tools/power.py).  Diagnostics only; nothing on the product path imports this.
    hipcc --offload-arch=gfx950 -O2 -shared -fPIC tools/probes/mfma_power_probe.hip -o /tmp/libmfma_power_probe.so
    python tools/mfma_power.py --lib /tmp/libmfma_power_probe.so --seconds 4 > gpurun_out/mfma_power.md
"""
import argparse
import ctypes
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from power import PowerSampler  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lib", default="/tmp/libmfma_power_probe.so")
    ap.add_argument("--seconds", type=float, default=4.0)
    ap.add_argument("--iters", type=int, default=40000)
    args = ap.parse_args()
    lib = ctypes.CDLL(args.lib)
    lib.mfma_power_launch.argtypes = [ctypes.c_int] * 5 + [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    dev = torch.device("cuda", 0)
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    g = torch.Generator(device=dev).manual_seed(0)
    n = cus * 512 * 16 * 8
    rnd = torch.randn(n, device=dev, generator=g).to(torch.bfloat16)
    zero = torch.zeros(n, device=dev, dtype=torch.bfloat16)
    sink = torch.zeros(4, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    sampler = PowerSampler(period=0.1)
    rows = []
    with sampler:
        time.sleep(2.0)
    idle = sampler.summary(skip_s=0.3)
    cases = [(shape, lds, waves, data) for data in ("random", "zero") for lds in (0, 1) for waves in (4, 8) for shape in (0, 1)]
    for shape, lds, waves, data in cases:
        buf = rnd if data == "random" else zero

        def launch():
            rc = lib.mfma_power_launch(shape, lds, waves, cus, args.iters, buf.data_ptr(), sink.data_ptr(), stream)
            assert rc == 0, rc
        launch()
        torch.cuda.synchronize()
        calls = 0
        with sampler:
            t0 = time.perf_counter()
            while time.perf_counter() - t0 < args.seconds:
                for _ in range(4):
                    launch()
                    calls += 1
                torch.cuda.synchronize()
            dt = time.perf_counter() - t0
        s = sampler.summary(skip_s=1.0)
        flop = 2.0 * 64 * 64 * 32 * args.iters * waves * cus * calls
        tf = flop / dt / 1e12
        # MFMA passes per second per SIMD against the clock: 4 cycles per pass
        passes = (16 * 4 if shape == 0 else 8 * 8) * args.iters * (waves / 4) * calls / dt
        rows.append((shape, lds, waves, data, tf, 1e3 * dt / calls, s, 4 * passes / (s.get("sclk_mhz", 0) * 1e6 + 1e-9)))
        print(rows[-1], file=sys.stderr, flush=True)
    print(f"idle: {idle.get('power_w', 0):.0f} W, {idle.get('sclk_mhz', 0):.0f} MHz ({idle.get('source')}); {cus} CUs, "
          f"{args.iters} block-steps of 64x64x32 per wave and launch, {args.seconds:.0f} s per case\n")
    print("| MFMA | operands | waves / SIMD | data | TF/s | ms / launch | package W | sclk MHz | matrix-pipe busy (passes x 4 / clock) | pJ / FLOP (package) | pJ / FLOP above idle |")
    print("|---|---|---|---|---|---|---|---|---|---|---|")
    for shape, lds, waves, data, tf, ms, s, busy in rows:
        w = s.get("power_w", 0.0)
        print(f"| {'16x16x32' if shape == 0 else '32x32x16'} | {'LDS (ds_read_b128 per step)' if lds else 'registers'} | {waves // 4} | {data} | "
              f"{tf:.0f} | {ms:.2f} | {w:.0f} | {s.get('sclk_mhz', 0):.0f} | {busy:.2f} | {w / tf:.3f} | {(w - idle.get('power_w', 0)) / tf:.3f} |")


if __name__ == "__main__":
    main()
