#!/bin/bash
# trip 12: grouped weight gradients: parity, chain bit-identity, in-step A/B
mkdir -p gpurun_out
export GPU_MAX_HW_QUEUES=8
timeout 900 python -m pytest tests/test_round3_gpu.py tests/test_round2_gpu.py -x -q -k "grouped or c_chain_is_bit_identical or options" 2>&1 | tail -15 > gpurun_out/r3t12_tests.log
echo "tests rc=$?" >> gpurun_out/r3t12_tests.log
timeout 600 python tools/abab.py --arms "base;nogroup:wgrad_group=0" --rounds 8 --steps 6 --power --out gpurun_out/r3t12_abab.json > gpurun_out/r3t12_abab.md 2> gpurun_out/r3t12_abab.err
echo "rc=$?" >> gpurun_out/r3t12_abab.md
