#!/bin/bash
# Round 5, trip 7: the epilogue's row operand (residual / saved gelu') requested up front (option gemm_epi_pre 1 / 2): bit-identity first,
# the isolated cost of the residual per shape, interleaved A/B in the step
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
(timeout 280 python -m pytest tests/test_round5_gpu.py -q -p no:cacheprovider -x -k "operand_preload" > $O/r5t7_tests.log 2>&1; echo "tests rc=$?" >> $O/r5t7_tests.log)
tail -5 $O/r5t7_tests.log
if ! grep -q "rc=0" $O/r5t7_tests.log; then grep -E "Error|error|assert" $O/r5t7_tests.log | head -20; echo "bit-identity failed: stopping"; exit 0; fi
(timeout 200 python tools/res_probe.py > $O/r5t7_res_probe.txt 2>&1); cat $O/r5t7_res_probe.txt
(timeout 500 python tools/abab.py --arms "base;pre1:gemm_epi_pre=1;pre2:gemm_epi_pre=2" --rounds 8 --steps 6 --out $O/r5t7_abab.json > $O/r5t7_abab.md 2> $O/r5t7_abab.err; echo "rc=$?" >> $O/r5t7_abab.err)
cat $O/r5t7_abab.md; tail -3 $O/r5t7_abab.err
