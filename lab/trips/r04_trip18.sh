#!/bin/bash
# Round 4, trip 18: the tree as it will be judged -- build check, whole GPU suite (with the emulation-parity tests), smoke, default bench line
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
(timeout 500 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/r4t18_bench.json 2> $O/r4t18_bench.err; echo "rc=$?" >> $O/r4t18_bench.err)
tail -2 $O/r4t18_bench.err | cut -c1-200
(timeout 1800 python -m pytest tests -x -q -m gpu -p no:cacheprovider > $O/r4t18_tests_all.log 2>&1; echo "tests rc=$?" >> $O/r4t18_tests_all.log)
grep -E "passed|failed|FAILED|ERROR|rc=" $O/r4t18_tests_all.log | tail -6
(timeout 200 python __graft_entry__.py --smoke > $O/r4t18_smoke.log 2>&1; echo "smoke rc=$?" >> $O/r4t18_smoke.log); tail -2 $O/r4t18_smoke.log
python - <<'PY'
import json
d=json.load(open('gpurun_out/r4t18_bench.json'))
print(d['value'], d['ms_per_step'], d['roofline']['frac'])
PY
