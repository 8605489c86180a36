#!/bin/bash
# Round 6, trip 1: baseline at HEAD (6 hardware queues now also under bench.py), the MFMA-filler probe the fused attention backward is
# designed on, the data-parallel line at one RCCL rank with the new fields, the GPU suite.
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
hipcc --offload-arch=gfx950 -O2 tools/probes/mfma_filler_probe.hip -o /tmp/mfma_filler_probe 2>/dev/null && timeout 300 /tmp/mfma_filler_probe > $O/r6t1_mfma_filler_probe.txt 2>&1
timeout 600 python bench.py --steps 20 --warmup 5 > $O/r6t1_bench.json 2> $O/r6t1_bench.err
tail -c 600 $O/r6t1_bench.json
timeout 300 env VJ_FORCE_DP=1 python bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline --no-roofline-pass > $O/r6t1_bench_dp1.json 2> $O/r6t1_bench_dp1.err
tail -c 1500 $O/r6t1_bench_dp1.json; tail -5 $O/r6t1_bench_dp1.err
timeout 200 python tools/attn_bench.py --reps 10 --shapes "prd ctx" --errors > $O/r6t1_attn_bench.txt 2>&1
cat $O/r6t1_attn_bench.txt
timeout 900 python -m pytest tests -x -q -m gpu > $O/r6t1_tests.txt 2>&1
tail -5 $O/r6t1_tests.txt
cat $O/r6t1_mfma_filler_probe.txt
