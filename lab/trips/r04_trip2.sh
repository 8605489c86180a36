#!/bin/bash
# Round 4, trip 2: attn_softmax = bias_fuse = 1 as defaults: the whole GPU suite, then the default bench line (roofline pass included)
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
(timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/r4t2_tests_all.log 2>&1; echo "tests rc=$?" >> $O/r4t2_tests_all.log)
grep -E "passed|failed|FAILED|ERROR|rc=" $O/r4t2_tests_all.log | tail -15
(timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --gemm-csv $O/r4t2_launches.csv > $O/r4t2_bench.json 2> $O/r4t2_bench.err; echo "rc=$?" >> $O/r4t2_bench.err)
tail -4 $O/r4t2_bench.err | cut -c1-400; cat $O/r4t2_bench.json | cut -c1-1500
