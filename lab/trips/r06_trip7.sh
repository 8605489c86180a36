#!/bin/bash
# Round 6, trip 7: joules per FLOP of the two MFMA shapes at the package cap (verdict item 6), the GPU suite with durations after the
# trim, the bench line.
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
hipcc --offload-arch=gfx950 -O2 -shared -fPIC tools/probes/mfma_power_probe.hip -o /tmp/libmfma_power_probe.so 2>$O/r6t7_probe_build.err \
  && timeout 400 python tools/mfma_power.py --lib /tmp/libmfma_power_probe.so --seconds 4 > $O/r6t7_mfma_power.md 2> $O/r6t7_mfma_power.err
cat $O/r6t7_mfma_power.md
( time timeout 1200 python -m pytest tests -x -q -m gpu --durations=40 ) > $O/r6t7_tests.txt 2>&1
tail -60 $O/r6t7_tests.txt
timeout 600 python bench.py --steps 20 --warmup 5 > $O/r6t7_bench.json 2> $O/r6t7_bench.err
tail -c 400 $O/r6t7_bench.json
