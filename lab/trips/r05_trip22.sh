#!/bin/bash
# Round 5, trip 22: the tree as committed on a fresh box, the way the driver runs it: bench line first, whole GPU suite, smoke; then the kernel
# statistics of the attention kernels again (they changed after trip 16) and the reducer path at one rank
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
(timeout 500 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/r5t22_bench.json 2> $O/r5t22_bench.err; echo "rc=$?" >> $O/r5t22_bench.err)
tail -2 $O/r5t22_bench.err | cut -c1-200; cut -c1-200 $O/r5t22_bench.json
(timeout 1800 python -m pytest tests -x -q -m gpu -p no:cacheprovider > $O/r5t22_tests_all.log 2>&1; echo "tests rc=$?" >> $O/r5t22_tests_all.log)
grep -E "passed|failed|FAILED|ERROR|rc=" $O/r5t22_tests_all.log | tail -5
(timeout 200 python __graft_entry__.py --smoke > $O/r5t22_smoke.log 2>&1; echo "smoke rc=$?" >> $O/r5t22_smoke.log); tail -2 $O/r5t22_smoke.log
cd /tmp
(VJ_NO_OVERLAP=1 timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof_r05c -o vjepa -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline-pass > $GRAFT_REPO_ROOT/$O/r5t22_prof_serial.log 2>&1)
cd $GRAFT_REPO_ROOT
db=$(find $O/prof_r05c -name "*results.db" | head -1); [ -n "$db" ] && python tools/rocpd_summary.py $db > $O/prof_r05c.md 2>/dev/null
find $O/prof_r05c -name "*.db" -delete
grep -E "attn_|persist_pre_kernel<0" $O/prof_r05c.md | head -12
(VJ_FORCE_DP=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline --no-roofline-pass > $O/r5t22_bench_dp1.json 2> $O/r5t22_bench_dp1.err; echo "rc=$?" >> $O/r5t22_bench_dp1.err); cut -c1-200 $O/r5t22_bench_dp1.json; tail -1 $O/r5t22_bench_dp1.err
(timeout 400 python tools/abab.py --arms "base;r4like:gemm_epi_pre=0,upd_overlap=0" --rounds 8 --steps 6 --out $O/r5t22_abab.json > $O/r5t22_abab.md 2> $O/r5t22_abab.err; echo "rc=$?" >> $O/r5t22_abab.err)
cat $O/r5t22_abab.md
