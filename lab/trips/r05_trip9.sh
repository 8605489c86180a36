#!/bin/bash
# Round 5, trip 9: are the epilogues of a launch burst-bound because its workgroups run in lockstep?  Staggered starts (option gemm_stagger),
# alone and with the dynamic tile hand-out: phase stamps per shape, per-shape rates, interleaved A/B in the step
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
for st in 0 4 8 16; do for dy in 0 1; do
(timeout 100 python tools/gemm_stamps.py 2 gemm_stagger=$st gemm_dyn=$dy >> $O/r5t9_stamps.txt 2>&1)
done; done
cat $O/r5t9_stamps.txt
(timeout 500 python tools/abab.py --arms "base;s4:gemm_stagger=4;s8:gemm_stagger=8;s16:gemm_stagger=16;s8d:gemm_stagger=8,gemm_dyn=1;s16d:gemm_stagger=16,gemm_dyn=1;pre3:gemm_epi_pre=3" --rounds 6 --steps 6 --out $O/r5t9_abab.json > $O/r5t9_abab.md 2> $O/r5t9_abab.err; echo "rc=$?" >> $O/r5t9_abab.err)
cat $O/r5t9_abab.md; tail -3 $O/r5t9_abab.err
