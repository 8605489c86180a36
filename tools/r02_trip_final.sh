#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out
(timeout 2000 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/r2f_tests.log 2>&1; echo "tests rc=$?" >> $O/r2f_tests.log)
(timeout 500 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/r2f_bench.json 2> $O/r2f_bench.err; echo "rc=$?" >> $O/r2f_bench.err)
grep -E "passed|failed|rc=" $O/r2f_tests.log | tail -3; tail -5 $O/r2f_bench.err | cut -c1-300
