#!/bin/bash
# Final evidence of the round: default bench on a fresh box FIRST (as the driver does), then the full GPU suite, the plumbing
# workload and a serial-view kernel trace.
export TMPDIR=/tmp
O=gpurun_out
(timeout 500 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/r2f_bench.json 2> $O/r2f_bench.err; echo "rc=$?" >> $O/r2f_bench.err)
(timeout 2000 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/r2f_tests.log 2>&1; echo "tests rc=$?" >> $O/r2f_tests.log)
(timeout 200 python bench.py --workload vittiny --steps 20 --warmup 5 --no-cpu-baseline > $O/r2f_tiny.json 2> $O/r2f_tiny.err; echo "rc=$?" >> $O/r2f_tiny.err)
cd /tmp
(VJ_NO_OVERLAP=1 timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof_r02f -o vjepa -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline-pass > $GRAFT_REPO_ROOT/$O/r2f_prof_serial.log 2>&1)
cd $GRAFT_REPO_ROOT
db=$(find $O/prof_r02f -name "*results.db" | head -1)
[ -n "$db" ] && python tools/rocpd_summary.py $db > $O/prof_r02f.md 2>/dev/null
find $O/prof_r02f -name "*.db" -size +40M -delete
grep -E "passed|failed|rc=" $O/r2f_tests.log | tail -3; tail -5 $O/r2f_bench.err | cut -c1-300; tail -2 $O/r2f_tiny.err | cut -c1-200; head -14 $O/prof_r02f.md
