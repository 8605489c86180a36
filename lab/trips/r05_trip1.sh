#!/bin/bash
# Round 5, trip 1: the new round-5 tests (deferred range-wise update, guard bands, persistent 4-wave GEMM), ABI test, isolated rates of
# the persistent 4-wave kernel, first interleaved A/B (update overlap, 4wp policies), default bench line
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
(timeout 900 python -m pytest tests/test_round5_gpu.py tests/test_abi.py -x -q -p no:cacheprovider > $O/r5t1_tests.log 2>&1; echo "tests rc=$?" >> $O/r5t1_tests.log)
tail -15 $O/r5t1_tests.log
(timeout 300 python tools/gemm_bench.py --no-wgrad --reps 20 --cfgs 8.4,9.0,4.0 > $O/r5t1_gemm_4wp.txt 2>&1); cat $O/r5t1_gemm_4wp.txt
(timeout 500 python tools/abab.py --arms "base:upd_overlap=0;upd;p4w3:gemm_4w=3;p4w4:gemm_4w=4;p4w5:gemm_4w=5" --rounds 6 --steps 6 --out $O/r5t1_abab.json > $O/r5t1_abab.md 2> $O/r5t1_abab.err; echo "rc=$?" >> $O/r5t1_abab.err)
cat $O/r5t1_abab.md; tail -3 $O/r5t1_abab.err
(timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/r5t1_bench.json 2> $O/r5t1_bench.err; echo "rc=$?" >> $O/r5t1_bench.err)
tail -3 $O/r5t1_bench.err | cut -c1-300; cut -c1-400 $O/r5t1_bench.json
