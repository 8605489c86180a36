#!/bin/bash
mkdir -p gpurun_out
export GPU_MAX_HW_QUEUES=8
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "attn or attention" 2>&1 | tail -4 > gpurun_out/r3t21_tests.log
timeout 300 python tools/attn_bench.py --reps 20 > gpurun_out/r3t21_attn_bench.log 2>&1
