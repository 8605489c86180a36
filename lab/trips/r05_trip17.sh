#!/bin/bash
# Round 5, trip 17: the reducer path at one RCCL rank was 19 % slower than the plain step in trip 16 (87.2 vs 73.2 ms).  Hypothesis: async_op=True
# collectives run on a pooled stream of ProcessGroupNCCL that shares a hardware queue with a compute stream.  Alternating processes: plain |
# reducer with async collectives | reducer with the collectives issued on the engine's own, checked communication stream
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
run() { # name env...
  local name=$1; shift
  (env "$@" timeout 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline --no-roofline-pass > $O/r5t17_$name.json 2> $O/r5t17_$name.err)
  python - <<PY
import json
d=json.loads(open("$O/r5t17_$name.json").read().strip().splitlines()[-1])
print("$name", d["value"], d["ms_per_step"], (d.get("dp") or {}).get("exposed_comm_ms_per_step"))
PY
  grep "stream picks" $O/r5t17_$name.err | cut -c1-400
}
for i in 1 2 3; do
run plain$i VJ_FORCE_DP=0
run async$i VJ_FORCE_DP=1 VJ_DP_COLL=async
run sync$i VJ_FORCE_DP=1 VJ_DP_COLL=sync
done
run sync_noupd VJ_FORCE_DP=1 VJ_DP_COLL=sync VJ_OVERLAP_UPDATE=0
run async_noupd VJ_FORCE_DP=1 VJ_DP_COLL=async VJ_OVERLAP_UPDATE=0
run sync_q16 VJ_FORCE_DP=1 VJ_DP_COLL=sync GPU_MAX_HW_QUEUES=16
run async_q16 VJ_FORCE_DP=1 VJ_DP_COLL=async GPU_MAX_HW_QUEUES=16
