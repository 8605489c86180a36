#!/bin/bash
# trip 17: strength-reduced issue path of the TN weight-gradient kernel
mkdir -p gpurun_out
export GPU_MAX_HW_QUEUES=8
timeout 900 python -m pytest tests/test_round3_gpu.py tests/test_kernels_gpu.py tests/test_round2_gpu.py -x -q -k "grouped or tn or wgrad or c_chain_is_bit_identical" 2>&1 | tail -5 > gpurun_out/r3t17_tests.log
timeout 200 python tools/wgrad_group_probe.py > gpurun_out/r3t17_probe.log 2>&1
VJ_WGRAD_SLOW_ISSUE=1 timeout 200 python tools/wgrad_group_probe.py > gpurun_out/r3t17_probe_slow.log 2>&1
timeout 900 python tools/abab.py --arms "base;slow:wgrad_slow_issue=1" --rounds 8 --steps 6 --out gpurun_out/r3t17_abab.json > gpurun_out/r3t17_abab.md 2> gpurun_out/r3t17_abab.err
echo "rc=$?" >> gpurun_out/r3t17_abab.md
