#!/bin/bash
# Round 5, trip 18: the reducer's collective modes at one RCCL rank -- torch.distributed on the engine's stream (sync, default), the C-ABI RCCL binding on
# the same stream (capi), ProcessGroupNCCL's own stream (async, the old form) -- and the two-ranks-on-one-GPU parity test with the new default
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
(timeout 500 python -m pytest tests/test_round2_gpu.py -q -p no:cacheprovider -x -k "rank or dp or reducer" > $O/r5t18_tests.log 2>&1; echo "tests rc=$?" >> $O/r5t18_tests.log); tail -3 $O/r5t18_tests.log
run() { # name env...
  local name=$1; shift
  (env "$@" timeout 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline --no-roofline-pass > $O/r5t18_$name.json 2> $O/r5t18_$name.err)
  python - <<PY
import json
try:
    d=json.loads(open("$O/r5t18_$name.json").read().strip().splitlines()[-1])
    print("$name", d["value"], d["ms_per_step"], (d.get("dp") or {}).get("exposed_comm_ms_per_step"), (d.get("dp") or {}).get("backend"))
except Exception as e:
    print("$name FAILED", e)
PY
  tail -2 $O/r5t18_$name.err | cut -c1-300
}
for i in 1 2; do
run plain$i VJ_FORCE_DP=0
run sync$i VJ_FORCE_DP=1 VJ_DP_COLL=sync
run capi$i VJ_FORCE_DP=1 VJ_DP_COLL=capi
run async$i VJ_FORCE_DP=1 VJ_DP_COLL=async
done
