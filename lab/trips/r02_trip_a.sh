#!/bin/bash
# Round-2 evidence trip: full GPU test suite, ViT-H workloads, kernel traces of the default bench (overlapped + serial).
export TMPDIR=/tmp
O=gpurun_out
(timeout 1500 python -m pytest tests -m gpu -q -x > $O/r2a_tests.log 2>&1; echo "tests rc=$?" >> $O/r2a_tests.log)
(timeout 400 python bench.py --workload vith16 --steps 3 --warmup 1 --no-cpu-baseline --gemm-csv $O/r2a_vith_gemm.csv > $O/r2a_vith16.json 2> $O/r2a_vith16.err; echo "rc=$?" >> $O/r2a_vith16.err)
(timeout 400 python bench.py --workload vith16_384 --steps 5 --warmup 2 --no-cpu-baseline --gemm-csv $O/r2a_vith384_gemm.csv > $O/r2a_vith16_384.json 2> $O/r2a_vith16_384.err; echo "rc=$?" >> $O/r2a_vith16_384.err)
(timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --h2d > $O/r2a_h2d.json 2> $O/r2a_h2d.err; echo "rc=$?" >> $O/r2a_h2d.err)
cd /tmp
(timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof_r02a -o vjepa -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline-pass > $GRAFT_REPO_ROOT/$O/r2a_prof_overlap.log 2>&1)
(VJ_NO_OVERLAP=1 timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof_r02b -o vjepa -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline-pass > $GRAFT_REPO_ROOT/$O/r2a_prof_serial.log 2>&1)
cd $GRAFT_REPO_ROOT
for d in prof_r02a prof_r02b; do
  db=$(find $O/$d -name "*results.db" | head -1)
  [ -n "$db" ] && python tools/rocpd_summary.py $db > $O/$d.md 2>/dev/null
  find $O/$d -name "*.db" -size +40M -delete
done
tail -3 $O/r2a_tests.log; tail -4 $O/r2a_vith16.err; tail -4 $O/r2a_vith16_384.err; tail -3 $O/r2a_h2d.err; head -12 $O/prof_r02b.md
