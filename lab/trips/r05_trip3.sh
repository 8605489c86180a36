#!/bin/bash
# Round 5, trip 3: whole GPU suite with the round-5 defaults (target LayerNorms folded, restructured attention backward, independent streams),
# hardware-queue aliasing probe, interleaved A/B of the fold and the deferred update, default bench line
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
(timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider -x > $O/r5t3_tests_all.log 2>&1; echo "tests rc=$?" >> $O/r5t3_tests_all.log)
grep -E "passed|failed|FAILED|ERROR|rc=|ln-fold|target fold" $O/r5t3_tests_all.log | tail -40
(timeout 300 python tools/queue_alias_probe.py --streams 20 --rounds 3 --steps 6 > $O/r5t3_queue_alias.md 2> $O/r5t3_queue_alias.err); cat $O/r5t3_queue_alias.md; tail -3 $O/r5t3_queue_alias.err
(timeout 600 python tools/abab.py --arms "base:ln_fold=0,upd_overlap=0;fold:upd_overlap=0;upd:ln_fold=0;both" --rounds 8 --steps 6 --out $O/r5t3_abab.json > $O/r5t3_abab.md 2> $O/r5t3_abab.err; echo "rc=$?" >> $O/r5t3_abab.err)
cat $O/r5t3_abab.md; tail -3 $O/r5t3_abab.err
(timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/r5t3_bench.json 2> $O/r5t3_bench.err; echo "rc=$?" >> $O/r5t3_bench.err)
tail -3 $O/r5t3_bench.err | cut -c1-300; cut -c1-300 $O/r5t3_bench.json
