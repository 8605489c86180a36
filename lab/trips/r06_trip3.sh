#!/bin/bash
# Round 6, trip 3: attention backward on MFMA 32x32x16 (head_dim <= 32): correctness against fp32 SDPA, speed against the 16x16x32 kernels
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -k "attention" > $O/r6t3_tests_a.txt 2>&1; tail -4 $O/r6t3_tests_a.txt
timeout 900 python -m pytest tests/test_round4_gpu.py -q -k "column_partials or several_segments or prescaled" > $O/r6t3_tests_b.txt 2>&1; tail -8 $O/r6t3_tests_b.txt
timeout 300 python tools/attn_bench.py --reps 10 --shapes "prd" --errors --opts "attn_mfma=16;attn_mfma=32;attn_mfma=32,attn_dkdv_kt=4;attn_mfma=16,attn_dkdv_kt=4" > $O/r6t3_attn_bench.txt 2>&1
grep -v "Warn\|amdgpu.ids\|rl = \|Consider" $O/r6t3_attn_bench.txt
