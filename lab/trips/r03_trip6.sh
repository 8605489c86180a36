#!/bin/bash
# Round 3, trip 6: restructured TN weight-gradient kernel (tests + isolated rates), low-priority weight-gradient stream A/B.
export TMPDIR=/tmp
O=gpurun_out
(timeout 900 python -m pytest tests/test_round3_gpu.py tests/test_kernels_gpu.py tests/test_round2_gpu.py tests/test_step_gpu.py -q -s -p no:cacheprovider -k "layernorm_bwd_column or wgrad_tn or c_chain or micro_batches or two_rank or vit_tiny or variance" > $O/r3t6_tests.log 2>&1; echo "tests rc=$?" >> $O/r3t6_tests.log)
grep -E "layernorm_bwd colsum|passed|failed|FAILED|ERROR|rc=" $O/r3t6_tests.log | tail -12
(timeout 200 python tools/wgrad_tn_bench.py > $O/r3t6_wgrad_tn.log 2>&1; echo "rc=$?" >> $O/r3t6_wgrad_tn.log)
cat $O/r3t6_wgrad_tn.log
(timeout 400 python tools/abab.py --power --rounds 8 --steps 5 --out $O/r3t6_abab.json --arms "base;lowprio:wgrad_lowprio=1;nt:wgrad_tn=0" > $O/r3t6_abab.md 2> $O/r3t6_abab.err; echo "rc=$?" >> $O/r3t6_abab.err)
cat $O/r3t6_abab.md; tail -2 $O/r3t6_abab.err
(VJ_PHASE_TIMING=1 timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-roofline-pass > $O/r3t6_bench.json 2> $O/r3t6_bench.err; echo "rc=$?" >> $O/r3t6_bench.err)
grep -E "timed region|phases" $O/r3t6_bench.err | cut -c1-600
