#!/bin/bash
# Round 6, trip 9: half tiles, second form (the idle late group issues both groups' LDS-DMA): bit-identity, interleaved per-shape A/B, step A/B
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
timeout 900 python -m pytest tests/test_gemm_gpu.py -x -q -m gpu > $O/r6t9_tests_gemm.txt 2>&1
tail -5 $O/r6t9_tests_gemm.txt
timeout 600 python tools/gemm_ab.py --option gemm_persist --values 3,1 --shapes 55680x384x384:res,55680x384x1536:res,55680x384x1152,58560x384x1536:res,52800x384x384:res,55680x1152x384,58560x1152x384,55680x1536x384:gelu,24576x640x512 > $O/r6t9_gemm_ab.md 2>&1
cat $O/r6t9_gemm_ab.md
timeout 900 python tools/abab.py --arms "full:gemm_persist=3;half:gemm_persist=1" --rounds 8 --steps 6 > $O/r6t9_abab.txt 2>&1
tail -4 $O/r6t9_abab.txt
timeout 900 python -m pytest tests/test_chain_gpu.py tests/test_step_gpu.py -x -q -m gpu > $O/r6t9_tests_step.txt 2>&1
tail -3 $O/r6t9_tests_step.txt
