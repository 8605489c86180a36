#!/bin/bash
# Round 4, trip 22: column groups of 4 / 6 and the K-dependent choice (511) once more, 8 rounds
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
(timeout 500 python tools/abab.py --arms "base;c4:gemm_raster=260;c6:gemm_raster=262;auto:gemm_raster=511" --rounds 8 --steps 6 --out $O/r4t22_abab.json > $O/r4t22_abab.md 2> $O/r4t22_abab.err; echo "rc=$?" >> $O/r4t22_abab.err)
cat $O/r4t22_abab.md; tail -2 $O/r4t22_abab.err
