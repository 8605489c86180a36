#!/bin/bash
# trip 25: fc1 epilogue saves gelu'(u) instead of u (backward epilogue = one multiply): parity tests, then the step
# with the previous library (VJ_LIB_VARIANT=base) and the new one, alternating on the same box
mkdir -p gpurun_out
export GPU_MAX_HW_QUEUES=8
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_round3_gpu.py -x -q -k "gemm or persist" 2>&1 | tail -4 > gpurun_out/r3t25_tests_kernels.log
timeout 900 python -m pytest tests/test_step_gpu.py -x -q 2>&1 | tail -4 > gpurun_out/r3t25_tests_step.log
for r in 1 2; do
  for v in base new; do
    if [ $v = base ]; then export VJ_LIB_VARIANT=base; else unset VJ_LIB_VARIANT; fi
    timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline-pass > gpurun_out/r3t25_bench_${v}_$r.json 2> gpurun_out/r3t25_bench_${v}_$r.err
  done
done
unset VJ_LIB_VARIANT
timeout 300 python tools/gemm_bench.py --reps 20 --cfgs 8.0 --no-wgrad > gpurun_out/r3t25_gemm_new.log 2>&1
