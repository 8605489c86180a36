#!/bin/bash
# Round 6, trip 11: the 4-wave 256x128 kernel (two workgroups per CU) for the PREDICTOR's GEMMs only -- its K = 384 GEMMs are epilogue-bound and its
# backward runs mostly alone on the chip, the regime in which that kernel won in isolation (+17 .. 35 %)
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
timeout 1200 python tools/abab.py --arms "base;pf:pred_flags=256;pb:pred_dgrad_flags=256;both:pred_flags=256,pred_dgrad_flags=256" --rounds 6 --steps 6 > $O/r6t11_abab.txt 2>&1
tail -8 $O/r6t11_abab.txt
timeout 600 python -m pytest tests/test_chain_gpu.py -x -q -m gpu > $O/r6t11_tests.txt 2>&1
tail -3 $O/r6t11_tests.txt
