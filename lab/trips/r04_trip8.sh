#!/bin/bash
# Round 4, trip 8: evidence at the current state -- default bench line (as the driver runs it), whole GPU suite, smoke, kernel traces
# (serial + two-stream), per-shape table, HBM / SQ counter passes (counters only, separate runs), the other two workloads
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
(timeout 500 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/r4t8_bench.json 2> $O/r4t8_bench.err; echo "rc=$?" >> $O/r4t8_bench.err)
tail -3 $O/r4t8_bench.err | cut -c1-300
(timeout 1800 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/r4t8_tests_all.log 2>&1; echo "tests rc=$?" >> $O/r4t8_tests_all.log)
grep -E "passed|failed|FAILED|ERROR|rc=" $O/r4t8_tests_all.log | tail -8
(timeout 200 python __graft_entry__.py --smoke > $O/r4t8_smoke.log 2>&1; echo "smoke rc=$?" >> $O/r4t8_smoke.log); tail -2 $O/r4t8_smoke.log
(timeout 300 python bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline --gemm-csv $O/r4t8_gemm.csv > $O/r4t8_bench_csv.json 2> $O/r4t8_bench_csv.err)
python tools/gemm_table.py $O/r4t8_gemm.csv 3 > $O/r4t8_gemm_shapes.md 2>&1
cd /tmp
(VJ_NO_OVERLAP=1 timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof_r04a -o vjepa -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline-pass > $GRAFT_REPO_ROOT/$O/r4t8_prof_serial.log 2>&1)
(timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof_r04b -o vjepa -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline-pass > $GRAFT_REPO_ROOT/$O/r4t8_prof_overlap.log 2>&1)
cd $GRAFT_REPO_ROOT
for v in a b; do db=$(find $O/prof_r04$v -name "*results.db" | head -1); [ -n "$db" ] && python tools/rocpd_summary.py $db > $O/prof_r04$v.md 2>/dev/null; done
find $O/prof_r04a $O/prof_r04b -name "*.db" -delete
head -30 $O/prof_r04a.md
mkdir -p $O/pmc_hbm_r04 $O/pmc_r04
bash tools/pmc_hbm.sh $O/pmc_hbm_r04 > $O/r4t8_pmc_hbm.log 2>&1
tail -30 $O/r4t8_pmc_hbm.log
bash tools/pmc_mfma.sh $O/pmc_r04 > $O/r4t8_pmc_mfma.log 2>&1
head -14 $O/pmc_r04/gemm_util.md; head -14 $O/pmc_r04/attn_util.md
find $O/pmc_hbm_r04 $O/pmc_r04 -name "*.csv" -size +8M -delete
(timeout 600 python bench.py --workload vith16 --steps 3 --warmup 1 --no-cpu-baseline > $O/r4t8_bench_vith16.json 2> $O/r4t8_bench_vith16.err; echo "rc=$?" >> $O/r4t8_bench_vith16.err); tail -2 $O/r4t8_bench_vith16.err | cut -c1-200
(timeout 400 python bench.py --workload vith16_384 --steps 8 --warmup 2 --no-cpu-baseline > $O/r4t8_bench_vith16_384.json 2> $O/r4t8_bench_vith16_384.err; echo "rc=$?" >> $O/r4t8_bench_vith16_384.err); tail -2 $O/r4t8_bench_vith16_384.err | cut -c1-200
