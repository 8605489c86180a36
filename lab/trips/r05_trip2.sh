#!/bin/bash
# Round 5, trip 2: guard-band tests with the corrected bookkeeping, restructured attention backward kernels (+ 64 keys / queries per wave),
# isolated attention rates, interleaved A/B: update overlap variants (priority, grid caps), attention tilings
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
(timeout 1200 python -m pytest tests/test_round5_gpu.py -q -p no:cacheprovider > $O/r5t2_tests5.log 2>&1; echo "tests rc=$?" >> $O/r5t2_tests5.log)
tail -25 $O/r5t2_tests5.log
(timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_round4_gpu.py -q -p no:cacheprovider -k "attn or attention or softmax or chain" > $O/r5t2_tests_attn.log 2>&1; echo "tests rc=$?" >> $O/r5t2_tests_attn.log)
tail -6 $O/r5t2_tests_attn.log
(timeout 300 python tools/attn_bench.py --reps 10 --shapes "ctx prd" --opts ";attn_dkdv_kt=4;attn_dq_qw=4;attn_dkdv_kt=4,attn_dq_qw=4" > $O/r5t2_attn_bench.txt 2>&1); cat $O/r5t2_attn_bench.txt
(timeout 700 python tools/abab.py --arms "base:upd_overlap=0;upd;updp:upd_prio=1;g1120:adam_grid=1120;g512:adam_grid=512;g256:adam_grid=256;kt4:attn_dkdv_kt=4;qw4:attn_dq_qw=4;kq4:attn_dkdv_kt=4,attn_dq_qw=4" --rounds 6 --steps 6 --out $O/r5t2_abab.json > $O/r5t2_abab.md 2> $O/r5t2_abab.err; echo "rc=$?" >> $O/r5t2_abab.err)
cat $O/r5t2_abab.md; tail -3 $O/r5t2_abab.err
