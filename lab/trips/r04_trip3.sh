#!/bin/bash
# Round 4, trip 3: the 4-section K-tile schedule of the persistent NT GEMM (option gemm_sched = 4): bit-identity, isolated GEMM table,
# interleaved step A/B; the round-4 tests again with their corrected bounds
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
(timeout 600 python -m pytest tests/test_round3_gpu.py -q -x -p no:cacheprovider -k persistent > $O/r4t3_persist.log 2>&1; echo "tests rc=$?" >> $O/r4t3_persist.log)
grep -E "passed|failed|FAILED|Error|rc=" $O/r4t3_persist.log | tail -6
(timeout 300 python tools/gemm_bench.py --reps 20 --cfgs 8.0,8.4 --no-wgrad > $O/r4t3_gemm.txt 2>&1; echo "rc=$?" >> $O/r4t3_gemm.txt)
cat $O/r4t3_gemm.txt | tail -24
(timeout 400 python tools/abab.py --arms "base;s4:gemm_sched=4" --rounds 6 --steps 6 --out $O/r4t3_abab.json > $O/r4t3_abab.md 2> $O/r4t3_abab.err; echo "rc=$?" >> $O/r4t3_abab.err)
cat $O/r4t3_abab.md; tail -2 $O/r4t3_abab.err
(timeout 900 python -m pytest tests/test_round4_gpu.py tests/test_round2_gpu.py -q -p no:cacheprovider -s -k "round4 or bit_identical or properties" > $O/r4t3_tests.log 2>&1; echo "tests rc=$?" >> $O/r4t3_tests.log)
grep -E "passed|failed|FAILED|Error|rc=|adversarial|worst" $O/r4t3_tests.log | tail -30
