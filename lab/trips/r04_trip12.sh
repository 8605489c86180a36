#!/bin/bash
# Round 4, trip 12: why do the 192-row dgrad tiles (+19..23 % isolated) not move the step?  Kernel timeline with the option off / on,
# solo time by kernel AND workgroup count (= shape), main-stream phase times for both
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
for v in 0 1; do
  (VJ_GEMM_BM192=$v VJ_PHASE_TIMING=1 timeout 300 python bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-roofline-pass > $O/r4t12_phase_bm$v.json 2> $O/r4t12_phase_bm$v.err)
  grep -E "main-stream phases|timed region" $O/r4t12_phase_bm$v.err | cut -c1-400
done
cd /tmp
for v in 0 1; do
  (VJ_GEMM_BM192=$v timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof_bm$v -o vjepa -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-roofline-pass > $GRAFT_REPO_ROOT/$O/r4t12_prof_bm$v.log 2>&1)
done
cd $GRAFT_REPO_ROOT
for v in 0 1; do db=$(find $O/prof_bm$v -name "*results.db" | head -1); if [ -n "$db" ]; then python tools/trace_timeline.py $db 3 > $O/timeline_bm$v.md 2>&1; fi; done
find $O/prof_bm0 $O/prof_bm1 -name "*.db" -delete
head -14 $O/timeline_bm0.md | tail -8
sed -n '/Time with exactly ONE/,$p' $O/timeline_bm0.md
echo "---- bm192 = 1"
head -14 $O/timeline_bm1.md | tail -8
sed -n '/Time with exactly ONE/,$p' $O/timeline_bm1.md
