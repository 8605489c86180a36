#!/bin/bash
# Round 3, trip 2: persistent 8-phase GEMM -- bit-identity tests first (own process, own timeout: a hang must not take the
# rest of the trip with it), isolated per-shape rates, K sweep, in-step interleaved A/B; then the whole GPU suite.
export TMPDIR=/tmp
O=gpurun_out
(timeout 420 python -m pytest tests/test_round3_gpu.py -q -x -p no:cacheprovider > $O/r3t2_tests_r3.log 2>&1; echo "tests rc=$?" >> $O/r3t2_tests_r3.log)
tail -5 $O/r3t2_tests_r3.log
(timeout 300 python tools/gemm_bench.py --no-wgrad --reps 20 --cfgs 2.3,8.0 > $O/r3t2_gemm_bench.log 2>&1; echo "rc=$?" >> $O/r3t2_gemm_bench.log)
cat $O/r3t2_gemm_bench.log
(timeout 300 python tools/gemm_ksweep.py > $O/r3t2_ksweep.log 2>&1; echo "rc=$?" >> $O/r3t2_ksweep.log)
grep -E "^---|slope" $O/r3t2_ksweep.log
(timeout 400 python tools/abab.py --power --rounds 8 --steps 5 --out $O/r3t2_abab.json --arms "base;persist:gemm_persist=1;nt:wgrad_tn=0" > $O/r3t2_abab.md 2> $O/r3t2_abab.err; echo "rc=$?" >> $O/r3t2_abab.err)
cat $O/r3t2_abab.md; tail -2 $O/r3t2_abab.err
(timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider -x --deselect tests/test_round3_gpu.py > $O/r3t2_tests_all.log 2>&1; echo "tests rc=$?" >> $O/r3t2_tests_all.log)
tail -15 $O/r3t2_tests_all.log
