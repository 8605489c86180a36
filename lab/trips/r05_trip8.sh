#!/bin/bash
# Round 5, trip 8: where a tile of the persistent GEMM spends its time (phase stamps, gemm_dbg = 4), the dGELU preload kernels without
# their spill (no-bias variant), option gemm_epi_pre = 3 (no blanket wait): bit-identity, per-shape cost, interleaved A/B in the step
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
(timeout 280 python -m pytest tests/test_round5_gpu.py -q -p no:cacheprovider -x -k "operand_preload" > $O/r5t8_tests.log 2>&1; echo "tests rc=$?" >> $O/r5t8_tests.log)
tail -5 $O/r5t8_tests.log
(timeout 200 python tools/gemm_stamps.py 2,0 > $O/r5t8_stamps.txt 2>&1); cat $O/r5t8_stamps.txt
(timeout 200 python tools/res_probe.py 0,2,3 > $O/r5t8_res_probe.txt 2>&1); cat $O/r5t8_res_probe.txt
if grep -q "rc=0" $O/r5t8_tests.log; then
(timeout 500 python tools/abab.py --arms "base;pre0:gemm_epi_pre=0;pre3:gemm_epi_pre=3" --rounds 6 --steps 6 --out $O/r5t8_abab.json > $O/r5t8_abab.md 2> $O/r5t8_abab.err; echo "rc=$?" >> $O/r5t8_abab.err)
cat $O/r5t8_abab.md; tail -3 $O/r5t8_abab.err
fi
