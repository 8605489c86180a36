#!/bin/bash
# Round 4, trip 15: HIP step vs the oracle with bf16 storage emulation (tests/test_emu_parity_gpu.py): measured errors
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
(timeout 600 python -m pytest tests/test_emu_parity_gpu.py -q -p no:cacheprovider -s > $O/r4t15_emu.log 2>&1; echo "tests rc=$?" >> $O/r4t15_emu.log)
grep -E "^\[|passed|failed|FAILED|Error|rc=|assert" $O/r4t15_emu.log | cut -c1-900 | tail -30
