#!/bin/bash
# Round 6, trip 4: software-pipelined dK/dV kernel on MFMA 32x32x16 (one / two waves per SIMD) against the phase-by-phase kernel and the 16x16x32 pair
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
run() { echo "=== $*"; env "$@" timeout 300 python tools/attn_bench.py --reps 10 --shapes "prd" --errors 2>&1 | grep -v "Warn\|amdgpu.ids\|rl = \|Consider"; }
run VJ_ATTN_MFMA=16 > $O/r6t4_attn_bench.txt
run VJ_ATTN_MFMA=32 VJ_ATTN_BWD32_PLAIN=1 >> $O/r6t4_attn_bench.txt
run VJ_ATTN_MFMA=32 >> $O/r6t4_attn_bench.txt
run VJ_ATTN_MFMA=32 VJ_LIB_VARIANT=lb2 >> $O/r6t4_attn_bench.txt
cat $O/r6t4_attn_bench.txt
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -k "attention_fwd_bwd" 2>&1 | tail -3
timeout 600 python -m pytest tests/test_round4_gpu.py -q -k "column_partials or several_segments or prescaled" 2>&1 | tail -3
