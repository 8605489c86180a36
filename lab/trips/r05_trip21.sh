#!/bin/bash
# Round 5, trip 21: attention kernels whose waves without rows (and halves without unpadded rows) skip the tile's arithmetic: every attention test,
# per-shape times against the previous library (variant "prev"), alternating bench processes
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
(timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider -x -k "attention or attn or softmax or segments or column_partials or prescaled" > $O/r5t21_tests.log 2>&1; echo "tests rc=$?" >> $O/r5t21_tests.log)
tail -3 $O/r5t21_tests.log
if ! grep -q "rc=0" $O/r5t21_tests.log; then grep -E "Error|error|assert|FAILED" $O/r5t21_tests.log | head -20; fi
SH="--shapes tgt,ctx"
(timeout 200 python tools/attn_bench.py --reps 10 > $O/r5t21_attn_new.txt 2>&1); (VJ_LIB_VARIANT=prev timeout 200 python tools/attn_bench.py --reps 10 > $O/r5t21_attn_prev.txt 2>&1)
echo "--- new"; cat $O/r5t21_attn_new.txt; echo "--- prev"; cat $O/r5t21_attn_prev.txt
for i in 1 2 3; do
(timeout 200 python bench.py --steps 15 --warmup 4 --no-cpu-baseline --no-roofline-pass 2>/dev/null | cut -c1-130 >> $O/r5t21_bench_new.txt)
(VJ_LIB_VARIANT=prev timeout 200 python bench.py --steps 15 --warmup 4 --no-cpu-baseline --no-roofline-pass 2>/dev/null | cut -c1-130 >> $O/r5t21_bench_prev.txt)
done
echo new; cat $O/r5t21_bench_new.txt; echo prev; cat $O/r5t21_bench_prev.txt
