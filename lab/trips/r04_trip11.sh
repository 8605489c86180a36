#!/bin/bash
# Round 4, trip 11: 192-row tiles of the persistent GEMM for the block chains' dgrad GEMMs (option gemm_bm192): bit identity (GEMM level
# and the ViT-L B=24 step), isolated context shapes, interleaved step A/B (0 = never, 1 = flagged dgrad launches (default), 2 = every launch)
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
(timeout 900 python -m pytest tests/test_round4_gpu.py -q -p no:cacheprovider -x -k "192" > $O/r4t11_tests.log 2>&1; echo "tests rc=$?" >> $O/r4t11_tests.log)
grep -E "passed|failed|FAILED|Error|rc=|assert" $O/r4t11_tests.log | tail -12
(timeout 200 python tools/gemm_bench.py --reps 20 --cfgs 8.4 --no-wgrad --toggle gemm_bm192=0,2 --only ctx > $O/r4t11_gemm.txt 2>&1; echo "rc=$?" >> $O/r4t11_gemm.txt)
cat $O/r4t11_gemm.txt | tail -9
(timeout 500 python tools/abab.py --power --arms "base;off:gemm_bm192=0;all:gemm_bm192=2" --rounds 6 --steps 6 --out $O/r4t11_abab.json > $O/r4t11_abab.md 2> $O/r4t11_abab.err; echo "rc=$?" >> $O/r4t11_abab.err)
cat $O/r4t11_abab.md; tail -2 $O/r4t11_abab.err
