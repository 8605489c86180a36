#!/bin/bash
# Round 5, trip 13: pipelined epilogue, running scalar row pointers for loads and stores, no-bias variants: bit-identity, phase stamps, per-shape rates, A/B in the step
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
(timeout 280 python -m pytest tests/test_round5_gpu.py -q -p no:cacheprovider -x -k "pipelined_epilogue or operand_preload" > $O/r5t13_tests.log 2>&1; echo "tests rc=$?" >> $O/r5t13_tests.log)
tail -5 $O/r5t13_tests.log
if ! grep -q "rc=0" $O/r5t13_tests.log; then grep -E "Error|error|assert" $O/r5t13_tests.log | head -20; fi
(timeout 200 python tools/gemm_stamps.py 4 > $O/r5t13_stamps.txt 2>&1); cat $O/r5t13_stamps.txt
(timeout 200 python tools/res_probe.py 2,3,4 > $O/r5t13_res_probe.txt 2>&1); cat $O/r5t13_res_probe.txt
(timeout 500 python tools/abab.py --arms "base;pre3:gemm_epi_pre=3;pre4:gemm_epi_pre=4" --rounds 8 --steps 6 --out $O/r5t13_abab.json > $O/r5t13_abab.md 2> $O/r5t13_abab.err; echo "rc=$?" >> $O/r5t13_abab.err)
cat $O/r5t13_abab.md; tail -3 $O/r5t13_abab.err
