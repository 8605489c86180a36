#!/bin/bash
# Round 5, trip 23: plain step and one-rank reducer (default collective mode, six hardware queues) in alternating fresh processes at HEAD
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
run() { local name=$1; shift
  (env "$@" timeout 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline --no-roofline-pass > $O/r5t23_$name.json 2> $O/r5t23_$name.err)
  python - <<PY
import json
d=json.loads(open("$O/r5t23_$name.json").read().strip().splitlines()[-1])
print("$name", d["value"], d["ms_per_step"], (d.get("dp") or {}).get("exposed_comm_ms_per_step"))
PY
}
for i in 1 2 3; do run plain$i VJ_FORCE_DP=0; run sync$i VJ_FORCE_DP=1; done
