#!/bin/bash
# Round 3, trip 5: LayerNorm-backward bias sums + faster colsum (tests, in-step A/B against the NT route), the read-skip
# timing experiment on the one-tile GEMM, bench.
export TMPDIR=/tmp
O=gpurun_out
(timeout 900 python -m pytest tests/test_round3_gpu.py tests/test_kernels_gpu.py tests/test_round2_gpu.py tests/test_step_gpu.py -q -x -p no:cacheprovider -k "not (eager or vit_huge or vit_large or full_size or ten_step)" > $O/r3t5_tests.log 2>&1; echo "tests rc=$?" >> $O/r3t5_tests.log)
grep -E "passed|failed|FAILED|ERROR|rc=" $O/r3t5_tests.log | tail -8
(VJ_LIB_VARIANT=skipreads timeout 200 python tools/gemm_ksweep.py > $O/r3t5_ksweep_skipreads.log 2>&1; echo "rc=$?" >> $O/r3t5_ksweep_skipreads.log)
grep -E "^===|^---|slope|K= 1024|K= 4096" $O/r3t5_ksweep_skipreads.log
(timeout 400 python tools/abab.py --power --rounds 8 --steps 5 --out $O/r3t5_abab.json --arms "base;nt:wgrad_tn=0;off:gemm_persist=0" > $O/r3t5_abab.md 2> $O/r3t5_abab.err; echo "rc=$?" >> $O/r3t5_abab.err)
cat $O/r3t5_abab.md; tail -2 $O/r3t5_abab.err
(timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/r3t5_bench.json 2> $O/r3t5_bench.err; echo "rc=$?" >> $O/r3t5_bench.err)
tail -3 $O/r3t5_bench.err | cut -c1-300
