#!/bin/bash
# trip 29: does confining the HBM-bound LayerNorm forward to a few CUs let the other stream's GEMMs run beside it?
mkdir -p gpurun_out
export GPU_MAX_HW_QUEUES=8
timeout 300 python -m pytest tests/test_kernels_gpu.py -x -q -k "layernorm" 2>&1 | tail -3 > gpurun_out/r3t29_tests.log
VJ_LN_FWD_CUS=64 timeout 300 python -m pytest tests/test_kernels_gpu.py -x -q -k "layernorm" 2>&1 | tail -3 >> gpurun_out/r3t29_tests.log
for c in 0 32 64 128; do VJ_LN_FWD_CUS=$c timeout 120 python tools/ln_bench.py 2>&1 | grep -v amdgpu | sed "s/^/cus=$c /" >> gpurun_out/r3t29_ln_bench.log; done
timeout 900 python tools/abab.py --arms "base;c32:ln_fwd_cus=32;c64:ln_fwd_cus=64;c128:ln_fwd_cus=128" --rounds 6 --steps 6 --out gpurun_out/r3t29_abab.json > gpurun_out/r3t29_abab.md 2> gpurun_out/r3t29_abab.err
