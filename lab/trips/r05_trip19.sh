#!/bin/bash
# Round 5, trip 19: what in the capi route stalls the compute streams at one rank although its communication stream is a checked one:
# the ncclAllReduce call itself (skipcall), the mere existence of a second communicator (extracomm), more hardware queues
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
run() { # name env...
  local name=$1; shift
  (env "$@" timeout 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline --no-roofline-pass > $O/r5t19_$name.json 2> $O/r5t19_$name.err)
  python - <<PY
import json
try:
    d=json.loads(open("$O/r5t19_$name.json").read().strip().splitlines()[-1])
    print("$name", d["value"], d["ms_per_step"], (d.get("dp") or {}).get("exposed_comm_ms_per_step"), (d.get("dp") or {}).get("backend"))
except Exception as e:
    print("$name FAILED", e)
PY
  grep "stream picks" $O/r5t19_$name.err | cut -c90-330
}
run sync VJ_FORCE_DP=1 VJ_DP_COLL=sync
run capi VJ_FORCE_DP=1 VJ_DP_COLL=capi
run capi_skipcall VJ_FORCE_DP=1 VJ_DP_COLL=capi VJ_DP_DIAG=skipcall
run sync_extracomm VJ_FORCE_DP=1 VJ_DP_COLL=sync VJ_DP_DIAG=extracomm
run capi_q24 VJ_FORCE_DP=1 VJ_DP_COLL=capi GPU_MAX_HW_QUEUES=24
run capi_q4 VJ_FORCE_DP=1 VJ_DP_COLL=capi GPU_MAX_HW_QUEUES=4
run capi_noupd VJ_FORCE_DP=1 VJ_DP_COLL=capi VJ_OVERLAP_UPDATE=0
