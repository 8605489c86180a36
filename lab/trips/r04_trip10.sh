#!/bin/bash
# Round 4, trip 10: default bench line with the exp2(polynomial) GELU, per-shape table, two-stream kernel trace + its TIMELINE analysis
# (tools/trace_timeline.py: idle / one-kernel / overlapped time per step, largest gaps)
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
(timeout 500 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/r4t10_bench.json 2> $O/r4t10_bench.err; echo "rc=$?" >> $O/r4t10_bench.err)
tail -3 $O/r4t10_bench.err | cut -c1-300
(timeout 300 python bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline --gemm-csv $O/r4t10_gemm.csv > $O/r4t10_bench_csv.json 2> $O/r4t10_bench_csv.err)
python tools/gemm_table.py $O/r4t10_gemm.csv 3 > $O/r4t10_gemm_shapes.md 2>&1
head -12 $O/r4t10_gemm_shapes.md
cd /tmp
(timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof_r04c -o vjepa -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-roofline-pass > $GRAFT_REPO_ROOT/$O/r4t10_prof_overlap.log 2>&1)
(VJ_NO_OVERLAP=1 timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof_r04d -o vjepa -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-roofline-pass > $GRAFT_REPO_ROOT/$O/r4t10_prof_serial.log 2>&1)
cd $GRAFT_REPO_ROOT
for v in c d; do db=$(find $O/prof_r04$v -name "*results.db" | head -1); if [ -n "$db" ]; then python tools/rocpd_summary.py $db > $O/prof_r04$v.md 2>/dev/null; python tools/trace_timeline.py $db 3 > $O/timeline_r04$v.md 2>&1; fi; done
find $O/prof_r04c $O/prof_r04d -name "*.db" -delete
cat $O/timeline_r04c.md
echo ---- serial
head -12 $O/timeline_r04d.md
