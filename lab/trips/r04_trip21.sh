#!/bin/bash
# Round 4, trip 21: column-grouped tile order, more group sizes and the automatic choice (gemm_raster = 511)
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
(timeout 500 python tools/abab.py --arms "base;c4:gemm_raster=260;c2:gemm_raster=258;c3:gemm_raster=259;c6:gemm_raster=262;auto:gemm_raster=511" --rounds 6 --steps 6 --out $O/r4t21_abab.json > $O/r4t21_abab.md 2> $O/r4t21_abab.err; echo "rc=$?" >> $O/r4t21_abab.err)
cat $O/r4t21_abab.md; tail -2 $O/r4t21_abab.err
(timeout 300 python tools/gemm_bench.py --reps 20 --cfgs 8.4 --no-wgrad --toggle gemm_raster=0,260,262,511 --only tgt,ctx,prd > $O/r4t21_gemm.txt 2>&1; echo "rc=$?" >> $O/r4t21_gemm.txt)
grep -v amdgpu.ids $O/r4t21_gemm.txt | cut -c1-160
