#!/bin/bash
# Round 3, trip 35: the final state once more after the dK/dV change -- default bench (as the driver runs it), whole GPU suite, smoke
export TMPDIR=/tmp
O=gpurun_out
(timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/r3t35_bench.json 2> $O/r3t35_bench.err; echo "rc=$?" >> $O/r3t35_bench.err)
tail -2 $O/r3t35_bench.err | cut -c1-300
(timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/r3t35_tests_all.log 2>&1; echo "tests rc=$?" >> $O/r3t35_tests_all.log)
grep -E "passed|failed|FAILED|ERROR|rc=" $O/r3t35_tests_all.log | tail -8
(timeout 200 python __graft_entry__.py --smoke > $O/r3t35_smoke.log 2>&1; echo "smoke rc=$?" >> $O/r3t35_smoke.log); tail -2 $O/r3t35_smoke.log
