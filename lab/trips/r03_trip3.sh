#!/bin/bash
# Round 3, trip 3: whole GPU suite (no -x: every failure at once) with the new defaults (persistent GEMM, TN weight gradients,
# fused attention delta, tightened gradient bounds), attention micro-bench, in-step A/B of the persistent grid policies, bench.
export TMPDIR=/tmp
O=gpurun_out
(timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/r3t3_tests_all.log 2>&1; echo "tests rc=$?" >> $O/r3t3_tests_all.log)
grep -E "passed|failed|FAILED|ERROR|rc=" $O/r3t3_tests_all.log | tail -25
(timeout 200 python tools/attn_bench.py > $O/r3t3_attn.log 2>&1; echo "rc=$?" >> $O/r3t3_attn.log)
cat $O/r3t3_attn.log | tail -16
(timeout 400 python tools/abab.py --power --rounds 8 --steps 5 --out $O/r3t3_abab.json --arms "base;full:gemm_persist=2;off:gemm_persist=0" > $O/r3t3_abab.md 2> $O/r3t3_abab.err; echo "rc=$?" >> $O/r3t3_abab.err)
cat $O/r3t3_abab.md; tail -2 $O/r3t3_abab.err
(timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/r3t3_bench.json 2> $O/r3t3_bench.err; echo "rc=$?" >> $O/r3t3_bench.err)
tail -3 $O/r3t3_bench.err | cut -c1-400
