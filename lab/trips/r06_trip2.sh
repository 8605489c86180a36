#!/bin/bash
# Round 6, trip 2: first run of the one-pass attention backward (correctness, then speed against the two-kernel form)
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
timeout 900 python -m pytest tests/test_attention_onepass_gpu.py -x -q > $O/r6t2_tests.txt 2>&1
tail -15 $O/r6t2_tests.txt
timeout 300 python tools/attn_bench.py --reps 10 --shapes "prd" --errors --opts "attn_bwd_fused=0;attn_bwd_fused=1" > $O/r6t2_attn_bench.txt 2>&1
cat $O/r6t2_attn_bench.txt
