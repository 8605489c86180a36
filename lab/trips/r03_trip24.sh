#!/bin/bash
# trip 24: 192-row tiles in the persistent GEMM: bit-identity tests, per-shape A/B, interleaved A/B in the step
mkdir -p gpurun_out
export GPU_MAX_HW_QUEUES=8
timeout 900 python -m pytest tests/test_round3_gpu.py -x -q -k "persist" 2>&1 | tail -6 > gpurun_out/r3t24_tests.log
for bm in 256 192 0; do
  VJ_GEMM_BM=$bm timeout 300 python tools/gemm_bench.py --reps 20 --cfgs 8.0 --no-wgrad > gpurun_out/r3t24_gemm_bm$bm.log 2>&1
done
timeout 900 python tools/abab.py --arms "bm256:gemm_bm=256;auto:gemm_bm=0" --rounds 8 --steps 6 --out gpurun_out/r3t24_abab.json > gpurun_out/r3t24_abab.md 2> gpurun_out/r3t24_abab.err
