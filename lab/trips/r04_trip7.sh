#!/bin/bash
# Round 4, trip 7: forward row sums from an all-ones MFMA operand (attn_psum for head sizes other than 24), q-column scale on q tiles only;
# attention + round-4 tests, attention micro-bench, interleaved step A/B
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
(timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_round4_gpu.py -q -p no:cacheprovider -k "attention or softmax or prescaled or qkv_gemm or segments or scale_in" > $O/r4t7_tests.log 2>&1; echo "tests rc=$?" >> $O/r4t7_tests.log)
grep -E "passed|failed|FAILED|Error|rc=" $O/r4t7_tests.log | tail -8
(timeout 200 python tools/attn_bench.py --reps 10 --only-fwd > $O/r4t7_attn.txt 2>&1; echo "rc=$?" >> $O/r4t7_attn.txt); tail -9 $O/r4t7_attn.txt
(timeout 400 python tools/abab.py --arms "base;nopsum:attn_psum=0;sm1:attn_softmax=1" --rounds 6 --steps 6 --out $O/r4t7_abab.json > $O/r4t7_abab.md 2> $O/r4t7_abab.err; echo "rc=$?" >> $O/r4t7_abab.err)
cat $O/r4t7_abab.md; tail -2 $O/r4t7_abab.err
