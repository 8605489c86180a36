#!/bin/bash
# Round 3, trip 8: evidence at the final state -- whole GPU suite, default bench (as the driver runs it), DP at one rank,
# serial + overlapped kernel traces, PMC passes (HBM traffic per kernel and per GEMM shape, MFMA utilisation), ViT-H lines.
export TMPDIR=/tmp
O=gpurun_out
(timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/r3t8_bench.json 2> $O/r3t8_bench.err; echo "rc=$?" >> $O/r3t8_bench.err)
tail -2 $O/r3t8_bench.err | cut -c1-300
(timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/r3t8_tests_all.log 2>&1; echo "tests rc=$?" >> $O/r3t8_tests_all.log)
grep -E "passed|failed|FAILED|ERROR|rc=" $O/r3t8_tests_all.log | tail -8
(timeout 200 python __graft_entry__.py --smoke > $O/r3t8_smoke.log 2>&1; echo "smoke rc=$?" >> $O/r3t8_smoke.log); tail -2 $O/r3t8_smoke.log
(VJ_FORCE_DP=1 timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-roofline-pass > $O/r3t8_dp1_torch.json 2> $O/r3t8_dp1_torch.err; echo "rc=$?" >> $O/r3t8_dp1_torch.err)
grep -E "exposed|timed" $O/r3t8_dp1_torch.err | cut -c1-200
(timeout 300 python bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline --gemm-csv $O/r3t8_gemm.csv > $O/r3t8_bench_csv.json 2> $O/r3t8_bench_csv.err)
python tools/gemm_table.py $O/r3t8_gemm.csv 3 > $O/r3t8_gemm_shapes.md 2>&1
cd /tmp
(VJ_NO_OVERLAP=1 timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof_r03f -o vjepa -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline-pass > $GRAFT_REPO_ROOT/$O/r3t8_prof_serial.log 2>&1)
(timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof_r03g -o vjepa -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline-pass > $GRAFT_REPO_ROOT/$O/r3t8_prof_overlap.log 2>&1)
cd $GRAFT_REPO_ROOT
for v in f g; do db=$(find $O/prof_r03$v -name "*results.db" | head -1); [ -n "$db" ] && python tools/rocpd_summary.py $db > $O/prof_r03$v.md 2>/dev/null; done
find $O/prof_r03f $O/prof_r03g -name "*.db" -delete
head -24 $O/prof_r03f.md
mkdir -p $O/pmc_hbm_r03 $O/pmc_r03
bash tools/pmc_hbm.sh $O/pmc_hbm_r03 > $O/r3t8_pmc_hbm.log 2>&1
tail -30 $O/r3t8_pmc_hbm.log
bash tools/pmc_mfma.sh $O/pmc_r03 > $O/r3t8_pmc_mfma.log 2>&1
head -12 $O/pmc_r03/gemm_util.md
find $O/pmc_hbm_r03 $O/pmc_r03 -name "*.csv" -size +8M -delete
(timeout 400 python bench.py --workload vith16 --steps 3 --warmup 1 --no-cpu-baseline > $O/r3t8_vith16.json 2> $O/r3t8_vith16.err; echo "rc=$?" >> $O/r3t8_vith16.err); grep -E "timed" $O/r3t8_vith16.err | cut -c1-200
(timeout 300 python bench.py --workload vith16_384 --steps 10 --warmup 3 --no-cpu-baseline > $O/r3t8_vith16_384.json 2> $O/r3t8_vith16_384.err; echo "rc=$?" >> $O/r3t8_vith16_384.err); grep -E "timed" $O/r3t8_vith16_384.err | cut -c1-200
(timeout 300 python bench.py --h2d --steps 20 --warmup 5 --no-cpu-baseline --no-roofline-pass > $O/r3t8_h2d.json 2> $O/r3t8_h2d.err); grep -E "input-edge" $O/r3t8_h2d.err | cut -c1-300
