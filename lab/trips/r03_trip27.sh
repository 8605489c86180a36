#!/bin/bash
# trip 27: plain instead of packed f32 arithmetic in the attention kernels (+ pipelined forward):
# parity / bit-identity tests, isolated base-vs-new, step base-vs-new alternating
mkdir -p gpurun_out
export GPU_MAX_HW_QUEUES=8
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "attention" 2>&1 | tail -4 > gpurun_out/r3t27_tests.log
for v in base new newold; do
  unset VJ_LIB_VARIANT VJ_ATTN_FWD_PIPE
  if [ $v = base ]; then export VJ_LIB_VARIANT=base; fi
  if [ $v = newold ]; then export VJ_ATTN_FWD_PIPE=0; fi
  timeout 300 python tools/attn_bench.py --reps 20 > gpurun_out/r3t27_attn_$v.log 2>&1
done
unset VJ_LIB_VARIANT VJ_ATTN_FWD_PIPE
for r in 1 2; do
  for v in base new; do
    if [ $v = base ]; then export VJ_LIB_VARIANT=base; else unset VJ_LIB_VARIANT; fi
    timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline-pass > gpurun_out/r3t27_bench_${v}_$r.json 2> gpurun_out/r3t27_bench_${v}_$r.err
  done
done
