#!/bin/bash
# Round 6, trip 13: dK/dV kernel with 32 keys per wave at head_dim 64 (context encoder S = 107 .. 366, target-size S = 1568 for ViT-H): isolated timing
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
for i in 1 2; do
echo "== KT=1 (shipped)"; timeout 200 python tools/attn_bench.py --reps 20 --shapes "tgt ctx" --errors
echo "== KT=2"; VJ_KT64=1 timeout 200 python tools/attn_bench.py --reps 20 --shapes "tgt ctx" --errors
done > $O/r6t13_attn_kt64.txt 2>&1
cat $O/r6t13_attn_kt64.txt
