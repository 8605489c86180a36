#!/bin/bash
# Round 4, trip 6: soft-max scale applied by the qkv GEMM epilogue (option attn_softmax = 2): kernel + chain tests, step A/B against option 1;
# kernel trace of the capi reducer at one rank (what runs on the GPU when vj_comm_* carries the buckets?)
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
(timeout 900 python -m pytest tests/test_round4_gpu.py -q -p no:cacheprovider -s -k "qkv_gemm or prescaled or scale_in_the_qkv or rebase" > $O/r4t6_tests.log 2>&1; echo "tests rc=$?" >> $O/r4t6_tests.log)
grep -E "passed|failed|FAILED|Error|rc=|adversarial|prescaled q" $O/r4t6_tests.log | tail -40
(timeout 400 python tools/abab.py --arms "sm1;sm2:attn_softmax=2" --rounds 6 --steps 6 --out $O/r4t6_abab.json > $O/r4t6_abab.md 2> $O/r4t6_abab.err; echo "rc=$?" >> $O/r4t6_abab.err)
cat $O/r4t6_abab.md; tail -2 $O/r4t6_abab.err
cd /tmp
(VJ_FORCE_DP=1 VJ_COMM_BACKEND=capi VJ_COMM_DEBUG=1 timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof_r04capi -o vjepa -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline-pass > $GRAFT_REPO_ROOT/$O/r4t6_capi_prof.log 2>&1)
cd $GRAFT_REPO_ROOT
db=$(find $O/prof_r04capi -name "*results.db" | head -1); [ -n "$db" ] && python tools/rocpd_summary.py $db > $O/prof_r04capi.md 2>/dev/null
find $O/prof_r04capi -name "*.db" -delete
grep -E "bound to|timed region|host enqueue" $O/r4t6_capi_prof.log | cut -c1-200
head -14 $O/prof_r04capi.md; grep -i -E "nccl|rccl|Kernel" $O/prof_r04capi.md | head
