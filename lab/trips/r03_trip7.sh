#!/bin/bash
# Round 3, trip 7: persistent GEMM with the fragment reads under the MFMAs (gemm8p2.hip): bit-identity, isolated rates, in-step A/B.
export TMPDIR=/tmp
O=gpurun_out
(timeout 600 python -m pytest tests/test_round3_gpu.py tests/test_step_gpu.py -q -x -p no:cacheprovider -k "persistent_gemm or variance" > $O/r3t7_tests.log 2>&1; echo "tests rc=$?" >> $O/r3t7_tests.log)
grep -E "passed|failed|FAILED|ERROR|rc=|assert" $O/r3t7_tests.log | tail -8
(timeout 300 python tools/gemm_bench.py --no-wgrad --reps 20 --cfgs 2.3,8.0,9.0 > $O/r3t7_gemm_bench.log 2>&1; echo "rc=$?" >> $O/r3t7_gemm_bench.log)
cat $O/r3t7_gemm_bench.log
(timeout 400 python tools/abab.py --power --rounds 8 --steps 5 --out $O/r3t7_abab.json --arms "base;p2:gemm_persist=3;off:gemm_persist=0" > $O/r3t7_abab.md 2> $O/r3t7_abab.err; echo "rc=$?" >> $O/r3t7_abab.err)
cat $O/r3t7_abab.md; tail -2 $O/r3t7_abab.err
