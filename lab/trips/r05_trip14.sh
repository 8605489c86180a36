#!/bin/bash
# Round 5, trip 14: the epilogue pipeline one pass deeper (gemm_epi_pre = 7) against the default (4): bit-identity, phase stamps, A/B in the step
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
(timeout 280 python -m pytest tests/test_round5_gpu.py -q -p no:cacheprovider -x -k "pipelined_epilogue" > $O/r5t14_tests.log 2>&1; echo "tests rc=$?" >> $O/r5t14_tests.log)
tail -3 $O/r5t14_tests.log
if ! grep -q "rc=0" $O/r5t14_tests.log; then grep -E "Error|error|assert" $O/r5t14_tests.log | head -20; fi
(timeout 200 python tools/gemm_stamps.py 7,4 > $O/r5t14_stamps.txt 2>&1); cat $O/r5t14_stamps.txt
(timeout 500 python tools/abab.py --arms "base;deep:gemm_epi_pre=7;pre2:gemm_epi_pre=2" --rounds 8 --steps 6 --out $O/r5t14_abab.json > $O/r5t14_abab.md 2> $O/r5t14_abab.err; echo "rc=$?" >> $O/r5t14_abab.err)
cat $O/r5t14_abab.md; tail -3 $O/r5t14_abab.err
