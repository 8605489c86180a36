#!/bin/bash
# Round 4, trip 17: the emulating oracle at ViT-L depth (GPU-eager, B = 8) against the HIP step
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
(timeout 900 python -m pytest tests/test_emu_parity_gpu.py -q -p no:cacheprovider -s -k vitl > $O/r4t17_emu_vitl.log 2>&1; echo "tests rc=$?" >> $O/r4t17_emu_vitl.log)
grep -E "^\[|passed|failed|FAILED|Error|rc=|assert|OutOfMemory" $O/r4t17_emu_vitl.log | cut -c1-900 | tail -12
