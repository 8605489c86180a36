#!/bin/bash
# Round 5, trip 11: what the ~3 us of a plain epilogue are made of: the pipelined epilogue (4) against its copies without global stores (5)
# and without the LDS round trip (6), phase stamps on the bias-only shapes
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
for pre in 4 5 6; do
(timeout 100 python tools/gemm_stamps.py $pre only=proj >> $O/r5t11_stamps.txt 2>&1)
(timeout 100 python tools/gemm_stamps.py $pre only=qkv >> $O/r5t11_stamps.txt 2>&1)
done
grep -v "res \|amdgpu" $O/r5t11_stamps.txt
