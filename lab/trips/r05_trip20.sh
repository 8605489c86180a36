#!/bin/bash
# Round 5, trip 20: number of hardware queues (GPU_MAX_HW_QUEUES) x {plain step, reducer via torch.distributed on the engine's stream, reducer via the C ABI}
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
run() { # name env...
  local name=$1; shift
  (env "$@" timeout 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline --no-roofline-pass > $O/r5t20_$name.json 2> $O/r5t20_$name.err)
  python - <<PY
import json
try:
    d=json.loads(open("$O/r5t20_$name.json").read().strip().splitlines()[-1])
    print("$name", d["value"], d["ms_per_step"], (d.get("dp") or {}).get("exposed_comm_ms_per_step"))
except Exception as e:
    print("$name FAILED", e)
PY
}
for q in 4 6 8 2; do
run plain_q$q VJ_FORCE_DP=0 GPU_MAX_HW_QUEUES=$q
run sync_q$q VJ_FORCE_DP=1 VJ_DP_COLL=sync GPU_MAX_HW_QUEUES=$q
run capi_q$q VJ_FORCE_DP=1 VJ_DP_COLL=capi GPU_MAX_HW_QUEUES=$q
done
