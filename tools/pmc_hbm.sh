#!/bin/bash
# HBM traffic passes (FETCH_SIZE / WRITE_SIZE in SEPARATE runs, counters only) for the step and for the per-shape audit.
set -u
OUT=${1:-gpurun_out/pmc_hbm_r02}
export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 400 rocprofv3 --pmc $c --output-format csv -d $OUT/step_$c -o s -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline-pass > $OUT.step_$c.log 2>&1
  timeout 300 rocprofv3 --pmc $c --output-format csv -d $OUT/shapes_$c -o s -- python tools/gemm_traffic.py run > $OUT.shapes_$c.log 2>&1
done
python tools/pmc_summary.py $OUT/step_FETCH_SIZE $OUT/step_WRITE_SIZE $OUT/step_hbm.md $OUT/step_hbm.json "HBM traffic per kernel of the ViT-L B=24 step (PMC)" > /dev/null
python tools/gemm_traffic.py summary $OUT/shapes_FETCH_SIZE $OUT/shapes_WRITE_SIZE $OUT/shapes_hbm.md | tail -24
head -24 $OUT/step_hbm.md
