#!/bin/bash
# Round 6, trip 8: half tiles in the persistent GEMM (N % 256 == 128): bit-identity, the predictor shapes in isolation, the step A/B
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
timeout 900 python -m pytest tests/test_gemm_gpu.py -x -q -m gpu > $O/r6t8_tests_gemm.txt 2>&1
tail -5 $O/r6t8_tests_gemm.txt
timeout 300 python tools/gemm_bench.py --reps 20 --no-wgrad --cfgs 8.0 --toggle gemm_persist=3,1 --only prd > $O/r6t8_gemm_bench.txt 2>&1
cat $O/r6t8_gemm_bench.txt
timeout 900 python tools/abab.py --arms "full:gemm_persist=3;half:gemm_persist=1" --rounds 8 --steps 6 > $O/r6t8_abab.txt 2>&1
tail -20 $O/r6t8_abab.txt
