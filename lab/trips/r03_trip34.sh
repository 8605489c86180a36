#!/bin/bash
# trip 34: 32 keys per wave in the attention dK/dV kernel: parity + bit-identity, isolated A/B, interleaved A/B in the step
mkdir -p gpurun_out
export GPU_MAX_HW_QUEUES=8
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "attention" 2>&1 | tail -4 > gpurun_out/r3t34_tests.log
for v in 1 2; do
  VJ_ATTN_DKDV_KT=$v timeout 300 python tools/attn_bench.py --reps 20 > gpurun_out/r3t34_attn_kt$v.log 2>&1
done
timeout 900 python tools/abab.py --arms "kt1:attn_dkdv_kt=1;auto:attn_dkdv_kt=0;kt2:attn_dkdv_kt=2" --rounds 6 --steps 6 --out gpurun_out/r3t34_abab.json > gpurun_out/r3t34_abab.md 2> gpurun_out/r3t34_abab.err
