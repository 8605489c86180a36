#!/bin/bash
# Round 3, trip 4: serial + overlapped kernel traces at the new defaults, per-shape GEMM table, printed parity figures.
export TMPDIR=/tmp
O=gpurun_out
(timeout 900 python -m pytest tests/test_step_gpu.py tests/test_round2_gpu.py -q -s -p no:cacheprovider -k "vit_tiny_step or variance_regulariser or oracle_run_by_eager or vit_huge_384 or vit_large_step_vs or vit_huge_step_vs" > $O/r3t4_parity.log 2>&1; echo "tests rc=$?" >> $O/r3t4_parity.log)
grep -E "gradient tensors|HIP loss|passed|failed|rc=" $O/r3t4_parity.log | cut -c1-600
(timeout 300 python bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline --gemm-csv $O/r3t4_gemm.csv > $O/r3t4_bench.json 2> $O/r3t4_bench.err; echo "rc=$?" >> $O/r3t4_bench.err)
python tools/gemm_table.py $O/r3t4_gemm.csv 3 > $O/r3t4_gemm_shapes.md 2>&1
head -50 $O/r3t4_gemm_shapes.md
cd /tmp
(VJ_NO_OVERLAP=1 timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof_r03a -o vjepa -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline-pass > $GRAFT_REPO_ROOT/$O/r3t4_prof_serial.log 2>&1)
(timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof_r03b -o vjepa -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline-pass > $GRAFT_REPO_ROOT/$O/r3t4_prof_overlap.log 2>&1)
cd $GRAFT_REPO_ROOT
for v in a b; do db=$(find $O/prof_r03$v -name "*results.db" | head -1); [ -n "$db" ] && python tools/rocpd_summary.py $db > $O/prof_r03$v.md 2>/dev/null; done
find $O/prof_r03a $O/prof_r03b -name "*.db" -size +40M -delete
head -45 $O/prof_r03a.md
