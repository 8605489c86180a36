#!/bin/bash
# trip 23: epilogue row-operand prefetch / cheaper GELU combination / -delta folded into the dP accumulators:
# kernel parity tests, then base-vs-new micro-benchmarks (VJ_LIB_VARIANT=base = the library built from HEAD~)
mkdir -p gpurun_out
export GPU_MAX_HW_QUEUES=8
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_round3_gpu.py -x -q -k "gemm or attn or attention or persist" 2>&1 | tail -6 > gpurun_out/r3t23_tests.log
for v in base new; do
  if [ $v = base ]; then export VJ_LIB_VARIANT=base; else unset VJ_LIB_VARIANT; fi
  timeout 200 python tools/res_probe.py > gpurun_out/r3t23_res_$v.log 2>&1
  timeout 200 python tools/attn_bench.py --reps 20 > gpurun_out/r3t23_attn_$v.log 2>&1
  timeout 300 python tools/gemm_bench.py --reps 20 --cfgs 8.0 --no-wgrad > gpurun_out/r3t23_gemm_$v.log 2>&1
done
