#!/bin/bash
# Round 5, trip 15: GELU epilogues evaluated in groups of independent chains (no hazard wait states): bit-identity, phase stamps, A/B against the
# previous library (trip 14's, kept as variant "t14")
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
(timeout 280 python -m pytest tests/test_round5_gpu.py tests/test_round4_gpu.py -q -p no:cacheprovider -x -k "pipelined_epilogue or gelu" > $O/r5t15_tests.log 2>&1; echo "tests rc=$?" >> $O/r5t15_tests.log)
tail -3 $O/r5t15_tests.log
if ! grep -q "rc=0" $O/r5t15_tests.log; then grep -E "Error|error|assert" $O/r5t15_tests.log | head -20; fi
(timeout 200 python tools/gemm_stamps.py 4 only=fc1 > $O/r5t15_stamps.txt 2>&1); cat $O/r5t15_stamps.txt
(VJ_LIB_VARIANT=t14 timeout 200 python tools/gemm_stamps.py 4 only=fc1 > $O/r5t15_stamps_t14.txt 2>&1); cat $O/r5t15_stamps_t14.txt
for i in 1 2 3; do
(timeout 200 python bench.py --steps 15 --warmup 4 --no-cpu-baseline --no-roofline-pass 2>/dev/null | cut -c1-120 >> $O/r5t15_bench_new.txt)
(VJ_LIB_VARIANT=t14 timeout 200 python bench.py --steps 15 --warmup 4 --no-cpu-baseline --no-roofline-pass 2>/dev/null | cut -c1-120 >> $O/r5t15_bench_t14.txt)
done
echo new; cat $O/r5t15_bench_new.txt; echo t14; cat $O/r5t15_bench_t14.txt
