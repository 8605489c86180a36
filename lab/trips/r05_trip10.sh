#!/bin/bash
# Round 5, trip 10: pipelined epilogue passes (gemm_epi_pre = 4): bit-identity, phase stamps, per-shape rates, interleaved A/B in the step
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
(timeout 280 python -m pytest tests/test_round5_gpu.py -q -p no:cacheprovider -x -k "pipelined_epilogue or operand_preload" > $O/r5t10_tests.log 2>&1; echo "tests rc=$?" >> $O/r5t10_tests.log)
tail -5 $O/r5t10_tests.log
if ! grep -q "rc=0" $O/r5t10_tests.log; then grep -E "Error|error|assert" $O/r5t10_tests.log | head -20; fi
(timeout 200 python tools/gemm_stamps.py 4,3 > $O/r5t10_stamps.txt 2>&1); cat $O/r5t10_stamps.txt
(timeout 200 python tools/res_probe.py 2,3,4 > $O/r5t10_res_probe.txt 2>&1); cat $O/r5t10_res_probe.txt
(timeout 200 python tools/gemm_bench.py --no-wgrad --reps 20 --cfgs 8.4 > $O/r5t10_gemm_pre2.txt 2>&1); (VJ_GEMM_EPI_PRE=4 timeout 200 python tools/gemm_bench.py --no-wgrad --reps 20 --cfgs 8.4 > $O/r5t10_gemm_pre4.txt 2>&1); paste $O/r5t10_gemm_pre2.txt $O/r5t10_gemm_pre4.txt | cut -c1-160
(timeout 500 python tools/abab.py --arms "base;pre3:gemm_epi_pre=3;pre4:gemm_epi_pre=4" --rounds 6 --steps 6 --out $O/r5t10_abab.json > $O/r5t10_abab.md 2> $O/r5t10_abab.err; echo "rc=$?" >> $O/r5t10_abab.err)
cat $O/r5t10_abab.md; tail -3 $O/r5t10_abab.err
