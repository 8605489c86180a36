#!/bin/bash
# Round 4, trip 16: AdamW kernel variants (option adam_variant): bit identity, isolated bandwidth, step A/B
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
(timeout 600 python -m pytest tests/test_round4_gpu.py tests/test_kernels_gpu.py -q -p no:cacheprovider -x -k "adamw" > $O/r4t16_tests.log 2>&1; echo "tests rc=$?" >> $O/r4t16_tests.log)
grep -E "passed|failed|FAILED|Error|rc=|assert" $O/r4t16_tests.log | tail -8
(timeout 200 python tools/adam_bench.py > $O/r4t16_adam_bench.txt 2>&1; echo "rc=$?" >> $O/r4t16_adam_bench.txt)
grep -v amdgpu.ids $O/r4t16_adam_bench.txt
(timeout 500 python tools/abab.py --arms "base;v1:adam_variant=1;v2:adam_variant=2" --rounds 8 --steps 6 --out $O/r4t16_abab.json > $O/r4t16_abab.md 2> $O/r4t16_abab.err; echo "rc=$?" >> $O/r4t16_abab.err)
cat $O/r4t16_abab.md; tail -2 $O/r4t16_abab.err
