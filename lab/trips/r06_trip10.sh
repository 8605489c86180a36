#!/bin/bash
# Round 6, trip 10: evidence at HEAD -- default bench line first (as the driver runs it), whole GPU suite, smoke, per-shape table, kernel traces
# (serial + two-stream) with timelines, HBM / SQ counter passes (counters only, separate runs; grouped weight gradients in the per-shape audit),
# the reducer at one rank, the other two workloads
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
(timeout 500 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/r6t10_bench.json 2> $O/r6t10_bench.err; echo "rc=$?" >> $O/r6t10_bench.err)
tail -3 $O/r6t10_bench.err | cut -c1-300; cut -c1-260 $O/r6t10_bench.json
(timeout 1800 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=25 > $O/r6t10_tests_all.log 2>&1; echo "tests rc=$?" >> $O/r6t10_tests_all.log)
grep -E "passed|failed|FAILED|ERROR|rc=" $O/r6t10_tests_all.log | tail -8
(timeout 200 python __graft_entry__.py --smoke > $O/r6t10_smoke.log 2>&1; echo "smoke rc=$?" >> $O/r6t10_smoke.log); tail -2 $O/r6t10_smoke.log
(timeout 300 python bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline --gemm-csv $O/r6t10_gemm.csv > $O/r6t10_bench_csv.json 2> $O/r6t10_bench_csv.err)
python tools/gemm_table.py $O/r6t10_gemm.csv 3 > $O/r6t10_gemm_shapes.md 2>&1; head -12 $O/r6t10_gemm_shapes.md
cd /tmp
(VJ_NO_OVERLAP=1 timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof_r06a -o vjepa -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline-pass > $GRAFT_REPO_ROOT/$O/r6t10_prof_serial.log 2>&1)
(timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof_r06b -o vjepa -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline-pass > $GRAFT_REPO_ROOT/$O/r6t10_prof_overlap.log 2>&1)
cd $GRAFT_REPO_ROOT
for v in a b; do db=$(find $O/prof_r06$v -name "*results.db" | head -1); [ -n "$db" ] && python tools/rocpd_summary.py $db > $O/prof_r06$v.md 2>/dev/null && python tools/trace_timeline.py $db 3 > $O/timeline_r06$v.md 2>/dev/null; done
find $O/prof_r06a $O/prof_r06b -name "*.db" -delete
head -24 $O/prof_r06a.md; head -30 $O/timeline_r06b.md
mkdir -p $O/pmc_hbm_r06 $O/pmc_r06
bash tools/pmc_hbm.sh $O/pmc_hbm_r06 > $O/r6t10_pmc_hbm.log 2>&1
tail -34 $O/r6t10_pmc_hbm.log
bash tools/pmc_mfma.sh $O/pmc_r06 > $O/r6t10_pmc_mfma.log 2>&1
head -14 $O/pmc_r06/gemm_util.md; head -14 $O/pmc_r06/attn_util.md
find $O/pmc_hbm_r06 $O/pmc_r06 -name "*.csv" -size +8M -delete
(VJ_FORCE_DP=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline > $O/r6t10_bench_dp1.json 2> $O/r6t10_bench_dp1.err; echo "rc=$?" >> $O/r6t10_bench_dp1.err); cut -c1-200 $O/r6t10_bench_dp1.json; tail -1 $O/r6t10_bench_dp1.err
(timeout 600 python bench.py --workload vith16 --steps 3 --warmup 1 --no-cpu-baseline > $O/r6t10_bench_vith16.json 2> $O/r6t10_bench_vith16.err; echo "rc=$?" >> $O/r6t10_bench_vith16.err); cut -c1-200 $O/r6t10_bench_vith16.json
(timeout 400 python bench.py --workload vith16_384 --steps 8 --warmup 2 --no-cpu-baseline > $O/r6t10_bench_vith16_384.json 2> $O/r6t10_bench_vith16_384.err; echo "rc=$?" >> $O/r6t10_bench_vith16_384.err); cut -c1-200 $O/r6t10_bench_vith16_384.json
