#!/bin/bash
# Round 6, trip 12: the committed tree as the driver runs it: bench line (dominant-kernel stats without the half-tile launches, round-6 traffic table),
# the GPU suite, smoke, and rocprofv3 --kernel-trace --stats of the same bench command
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
(timeout 500 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/r6t12_bench.json 2> $O/r6t12_bench.err; echo "rc=$?" >> $O/r6t12_bench.err)
tail -2 $O/r6t12_bench.err | cut -c1-200; cut -c1-200 $O/r6t12_bench.json
(time timeout 1500 python -m pytest tests -x -q -m gpu --durations=15 ) > $O/r6t12_tests.txt 2>&1
tail -25 $O/r6t12_tests.txt
(timeout 200 python __graft_entry__.py --smoke > $O/r6t12_smoke.log 2>&1; echo "smoke rc=$?" >> $O/r6t12_smoke.log); tail -2 $O/r6t12_smoke.log
cd /tmp
(timeout 400 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof_r06c -o vjepa -- python $GRAFT_REPO_ROOT/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/r6t12_prof.log 2>&1)
cd $GRAFT_REPO_ROOT
db=$(find $O/prof_r06c -name "*results.db" | head -1); [ -n "$db" ] && python tools/rocpd_summary.py $db > $O/prof_r06c.md 2>/dev/null
find $O/prof_r06c -name "*.db" -delete
head -14 $O/prof_r06c.md | cut -c1-150
tail -c 700 $O/r6t12_prof.log
