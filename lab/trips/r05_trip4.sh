#!/bin/bash
# Round 5, trip 4: dynamic tile hand-out of the persistent GEMM (bit-identity first, under a short timeout), the corrected folded-LayerNorm
# epilogue, the step test that failed in trip 3 (fold now opt-in), isolated GEMM rates dyn on / off, interleaved A/B
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
(timeout 300 python -m pytest tests/test_round5_gpu.py -q -p no:cacheprovider -x -k "dynamic or fold or folded" -s > $O/r5t4_tests_dyn.log 2>&1; echo "tests rc=$?" >> $O/r5t4_tests_dyn.log)
grep -E "passed|failed|FAILED|ERROR|rc=|ln-fold|target fold|folded target" $O/r5t4_tests_dyn.log | tail -50
if grep -q "rc=124" $O/r5t4_tests_dyn.log; then echo "TIMEOUT in the first test block: stopping the trip"; exit 0; fi
(timeout 900 python -m pytest tests/test_step_gpu.py tests/test_train_loop_gpu.py tests/test_round3_gpu.py -q -p no:cacheprovider -k "not vith" > $O/r5t4_tests_step.log 2>&1; echo "tests rc=$?" >> $O/r5t4_tests_step.log)
tail -4 $O/r5t4_tests_step.log
(timeout 300 python tools/gemm_bench.py --no-wgrad --reps 20 --cfgs 8.4 --toggle gemm_dyn > $O/r5t4_gemm_dyn.txt 2>&1); cat $O/r5t4_gemm_dyn.txt
(timeout 700 python tools/abab.py --arms "base;dyn:gemm_dyn=1;dynfull:gemm_dyn=1,gemm_persist=2;full:gemm_persist=2;fold:ln_fold=1;folddyn:ln_fold=1,gemm_dyn=1" --rounds 8 --steps 6 --out $O/r5t4_abab.json > $O/r5t4_abab.md 2> $O/r5t4_abab.err; echo "rc=$?" >> $O/r5t4_abab.err)
cat $O/r5t4_abab.md; tail -3 $O/r5t4_abab.err
