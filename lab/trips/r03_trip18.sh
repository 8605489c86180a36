#!/bin/bash
mkdir -p gpurun_out
export GPU_MAX_HW_QUEUES=8
timeout 900 python tools/abab.py --arms "base;p3:gemm_persist=3" --rounds 6 --steps 6 --out gpurun_out/r3t18_abab.json > gpurun_out/r3t18_abab.md 2> gpurun_out/r3t18_abab.err
echo "rc=$?" >> gpurun_out/r3t18_abab.md
