#!/bin/bash
# Round 3, trip 32: evidence at the final state -- default bench (as the driver runs it), whole GPU suite, smoke, DP at one rank,
# per-shape GEMM table, serial + overlapped kernel traces, the ViT-H lines
export TMPDIR=/tmp
O=gpurun_out
(timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/r3t32_bench.json 2> $O/r3t32_bench.err; echo "rc=$?" >> $O/r3t32_bench.err)
tail -2 $O/r3t32_bench.err | cut -c1-300
(timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/r3t32_tests_all.log 2>&1; echo "tests rc=$?" >> $O/r3t32_tests_all.log)
grep -E "passed|failed|FAILED|ERROR|rc=" $O/r3t32_tests_all.log | tail -8
(timeout 200 python __graft_entry__.py --smoke > $O/r3t32_smoke.log 2>&1; echo "smoke rc=$?" >> $O/r3t32_smoke.log); tail -2 $O/r3t32_smoke.log
(VJ_FORCE_DP=1 GPU_MAX_HW_QUEUES=8 timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-roofline-pass > $O/r3t32_dp1_torch.json 2> $O/r3t32_dp1_torch.err; echo "rc=$?" >> $O/r3t32_dp1_torch.err)
grep -E "exposed|timed" $O/r3t32_dp1_torch.err | cut -c1-200
(timeout 300 python bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline --gemm-csv $O/r3t32_gemm.csv > $O/r3t32_bench_csv.json 2> $O/r3t32_bench_csv.err)
python tools/gemm_table.py $O/r3t32_gemm.csv 3 > $O/r3t32_gemm_shapes.md 2>&1
cd /tmp
(VJ_NO_OVERLAP=1 timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof_r03h -o vjepa -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline-pass > $GRAFT_REPO_ROOT/$O/r3t32_prof_serial.log 2>&1)
(timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof_r03i -o vjepa -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline-pass > $GRAFT_REPO_ROOT/$O/r3t32_prof_overlap.log 2>&1)
cd $GRAFT_REPO_ROOT
for v in h i; do db=$(find $O/prof_r03$v -name "*results.db" | head -1); [ -n "$db" ] && python tools/rocpd_summary.py $db > $O/prof_r03$v.md 2>/dev/null; done
find $O/prof_r03h $O/prof_r03i -name "*.db" -delete
head -16 $O/prof_r03h.md
(timeout 300 python bench.py --workload vith16_384 --steps 10 --warmup 3 --no-cpu-baseline > $O/r3t32_vith16_384.json 2> $O/r3t32_vith16_384.err; echo "rc=$?" >> $O/r3t32_vith16_384.err); grep -E "timed" $O/r3t32_vith16_384.err | cut -c1-200
(timeout 400 python bench.py --workload vith16 --steps 3 --warmup 1 --no-cpu-baseline > $O/r3t32_vith16.json 2> $O/r3t32_vith16.err; echo "rc=$?" >> $O/r3t32_vith16.err); grep -E "timed" $O/r3t32_vith16.err | cut -c1-200
