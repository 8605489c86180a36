#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out
(timeout 1800 python -m pytest tests -m gpu -q > $O/r2b_tests.log 2>&1; echo "tests rc=$?" >> $O/r2b_tests.log)
(timeout 300 python __graft_entry__.py --smoke > $O/r2b_smoke.log 2>&1; echo "smoke rc=$?" >> $O/r2b_smoke.log)
for be in torch capi; do
  (VJ_FORCE_DP=1 VJ_COMM_BACKEND=$be timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29544 bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline --no-roofline-pass > $O/r2b_dp1_$be.json 2> $O/r2b_dp1_$be.err; echo "rc=$?" >> $O/r2b_dp1_$be.err)
done
tail -4 $O/r2b_tests.log; tail -2 $O/r2b_smoke.log; for be in torch capi; do grep -E "data parallel|exposed|timed|rc=" $O/r2b_dp1_$be.err; done
