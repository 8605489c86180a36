#!/bin/bash
# Round 5, trip 24: package power and shader clock per case (tools/power_matrix.py) -- is the s_memtime rate of the phase stamps (1.6 ticks/ns in
# back-to-back launches of a step GEMM) the shader clock?
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
(timeout 400 python tools/power_matrix.py --seconds 3 > $O/r5t24_power_matrix.md 2> $O/r5t24_power_matrix.err); cat $O/r5t24_power_matrix.md; tail -3 $O/r5t24_power_matrix.err | cut -c1-300
(timeout 100 python tools/gemm_stamps.py 4 only=qkv > $O/r5t24_stamps.txt 2>&1); cat $O/r5t24_stamps.txt
