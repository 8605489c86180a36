#!/bin/bash
mkdir -p gpurun_out
export GPU_MAX_HW_QUEUES=8
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_round3_gpu.py -x -q -k "gemm" 2>&1 | tail -5 > gpurun_out/r3t19_tests.log
timeout 900 python tools/abab.py --arms "base;n384:gemm_4w=2;p0:gemm_persist=0" --rounds 6 --steps 6 --out gpurun_out/r3t19_abab.json > gpurun_out/r3t19_abab.md 2> gpurun_out/r3t19_abab.err
echo "rc=$?" >> gpurun_out/r3t19_abab.md
