#!/bin/bash
# Round 4, trip 5: attention over all segments of a block in one launch (option attn_merge): bit-identity tests, interleaved step A/B;
# the data-parallel reducer at one RCCL rank with both collective back ends (plain / torch.distributed / vj_comm_*)
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
(timeout 600 python -m pytest tests/test_round4_gpu.py -q -p no:cacheprovider -k "segments or chain" > $O/r4t5_tests.log 2>&1; echo "tests rc=$?" >> $O/r4t5_tests.log)
grep -E "passed|failed|FAILED|Error|rc=" $O/r4t5_tests.log | tail -8
(timeout 400 python tools/abab.py --arms "base;nomerge:attn_merge=0" --rounds 6 --steps 6 --out $O/r4t5_abab.json > $O/r4t5_abab.md 2> $O/r4t5_abab.err; echo "rc=$?" >> $O/r4t5_abab.err)
cat $O/r4t5_abab.md; tail -2 $O/r4t5_abab.err
for mode in plain torch capi; do
  if [ $mode = plain ]; then E=""; elif [ $mode = torch ]; then E="VJ_FORCE_DP=1"; else E="VJ_FORCE_DP=1 VJ_COMM_BACKEND=capi"; fi
  (env $E timeout 300 python bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-roofline-pass > $O/r4t5_dp1_$mode.json 2> $O/r4t5_dp1_$mode.err; echo "rc=$?" >> $O/r4t5_dp1_$mode.err)
  echo "== $mode"; grep -E "host enqueue|exposed|timed region|rc=" $O/r4t5_dp1_$mode.err | cut -c1-300
done
