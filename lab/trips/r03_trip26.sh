#!/bin/bash
# trip 26: software-pipelined attention forward: bit-identity + parity tests, isolated A/B, interleaved A/B in the step
mkdir -p gpurun_out
export GPU_MAX_HW_QUEUES=8
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "attention" 2>&1 | tail -6 > gpurun_out/r3t26_tests.log
for v in 0 1; do
  VJ_ATTN_FWD_PIPE=$v timeout 300 python tools/attn_bench.py --reps 20 --only-fwd > gpurun_out/r3t26_attn_pipe$v.log 2>&1
done
timeout 900 python tools/abab.py --arms "old:attn_fwd_pipe=0;pipe:attn_fwd_pipe=1" --rounds 6 --steps 6 --out gpurun_out/r3t26_abab.json > gpurun_out/r3t26_abab.md 2> gpurun_out/r3t26_abab.err
