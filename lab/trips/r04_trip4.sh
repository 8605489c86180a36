#!/bin/bash
# Round 4, trip 4: gemm_sched = 4 default; target-encoder GEMM selection (4-wave two-workgroups-per-CU kernel for all / late blocks) in the
# interleaved step A/B; head_dim-24 pad-column row sums vs vector row sums (accuracy + speed); main-stream phase timing
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
(timeout 600 python -m pytest tests/test_round4_gpu.py -q -p no:cacheprovider -s -k "rebase" > $O/r4t4_tests.log 2>&1; echo "tests rc=$?" >> $O/r4t4_tests.log)
grep -E "passed|failed|FAILED|Error|rc=|adversarial" $O/r4t4_tests.log | tail -20
# 0x100 = 256 (all blocks); from block 8: 256 + 8*65536 = 524544; from 12: 786688; from 16: 1048832
(timeout 500 python tools/abab.py --arms "base;t4w:tgt_flags=256;t4w8:tgt_flags=524544;t4w12:tgt_flags=786688;t4w16:tgt_flags=1048832;nopsum:attn_psum=0" --rounds 4 --steps 6 --out $O/r4t4_abab.json > $O/r4t4_abab.md 2> $O/r4t4_abab.err; echo "rc=$?" >> $O/r4t4_abab.err)
cat $O/r4t4_abab.md; tail -2 $O/r4t4_abab.err
(VJ_PHASE_TIMING=1 timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline-pass > $O/r4t4_bench.json 2> $O/r4t4_bench.err; echo "rc=$?" >> $O/r4t4_bench.err)
grep -E "phases|timed region|rc=" $O/r4t4_bench.err | cut -c1-600
