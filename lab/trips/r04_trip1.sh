#!/bin/bash
# Round 4, trip 1: new kernels behind options (attn_softmax, bias_fuse): parity tests, attention A/B with errors, interleaved step A/B,
# the --gpus 2 launch rehearsal on a 1-GPU box
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
(timeout 900 python -m pytest tests/test_round4_gpu.py -q -x -p no:cacheprovider -s > $O/r4t1_tests.log 2>&1; echo "tests rc=$?" >> $O/r4t1_tests.log)
grep -E "passed|failed|FAILED|Error|rc=|worst" $O/r4t1_tests.log | tail -12
(timeout 300 python tools/attn_bench.py --reps 10 --sm 0,1 --errors > $O/r4t1_attn.txt 2>&1; echo "rc=$?" >> $O/r4t1_attn.txt)
cat $O/r4t1_attn.txt | tail -20
(timeout 400 python tools/abab.py --arms "base;sm:attn_softmax=1;bf:bias_fuse=1;both:attn_softmax=1,bias_fuse=1" --rounds 4 --steps 6 --out $O/r4t1_abab.json > $O/r4t1_abab.md 2> $O/r4t1_abab.err; echo "rc=$?" >> $O/r4t1_abab.err)
cat $O/r4t1_abab.md; tail -3 $O/r4t1_abab.err
(timeout 120 python bench.py --gpus 2 --steps 3 --warmup 1 > $O/r4t1_gpus2.out 2> $O/r4t1_gpus2.err; echo "rc=$?" >> $O/r4t1_gpus2.err)
tail -2 $O/r4t1_gpus2.err
