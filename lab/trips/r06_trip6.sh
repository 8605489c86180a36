#!/bin/bash
# Round 6, trip 6: after the prune (12 options, .so 2.4 MB) and the regrouping of the tests: the whole GPU suite, smoke, the bench line
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
( time timeout 1500 python -m pytest tests -q -m gpu --durations=15 ) > $O/r6t6_tests.txt 2>&1
tail -40 $O/r6t6_tests.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/r6t6_bench.json 2> $O/r6t6_bench.err
python - <<PY
import json
d=json.loads(open("$O/r6t6_bench.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["attn_bwd"], d["roofline"]["gemm_family"]["achieved"])
PY
