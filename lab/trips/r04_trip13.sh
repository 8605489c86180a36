#!/bin/bash
# Round 4, trip 13: LayerNorm backward with a one-row-ahead software prefetch (option ln_bwd_prefetch): tests, isolated bandwidth, step A/B
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
(timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_round3_gpu.py -q -p no:cacheprovider -x -k "layernorm or colsum or ln_" > $O/r4t13_tests.log 2>&1; echo "tests rc=$?" >> $O/r4t13_tests.log)
grep -E "passed|failed|FAILED|Error|rc=|assert" $O/r4t13_tests.log | tail -8
(timeout 200 python tools/ln_bench.py > $O/r4t13_ln_bench.txt 2>&1; echo "rc=$?" >> $O/r4t13_ln_bench.txt)
grep -v amdgpu.ids $O/r4t13_ln_bench.txt
(timeout 500 python tools/abab.py --arms "base;nopf:ln_bwd_prefetch=0" --rounds 8 --steps 6 --out $O/r4t13_abab.json > $O/r4t13_abab.md 2> $O/r4t13_abab.err; echo "rc=$?" >> $O/r4t13_abab.err)
cat $O/r4t13_abab.md; tail -2 $O/r4t13_abab.err
(timeout 900 python -m pytest tests/test_step_gpu.py -q -p no:cacheprovider -x > $O/r4t13_tests_b.log 2>&1; echo "tests rc=$?" >> $O/r4t13_tests_b.log)
grep -E "passed|failed|FAILED|Error|rc=" $O/r4t13_tests_b.log | tail -5
