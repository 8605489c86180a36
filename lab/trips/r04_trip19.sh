#!/bin/bash
# Round 4, trip 19: EXPERIMENTAL one-wave-per-SIMD GEMM (csrc/gemm1w.hip): correctness, K-sweep against the production kernels
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
(timeout 300 python -m pytest tests/test_round4_gpu.py -q -p no:cacheprovider -x -k "one_wave" > $O/r4t19_tests.log 2>&1; echo "tests rc=$?" >> $O/r4t19_tests.log)
grep -E "passed|failed|FAILED|Error|rc=|assert" $O/r4t19_tests.log | tail -8
(timeout 400 python tools/gemm1w_bench.py > $O/r4t19_gemm1w.txt 2>&1; echo "rc=$?" >> $O/r4t19_gemm1w.txt)
grep -v amdgpu.ids $O/r4t19_gemm1w.txt | cut -c1-330
