#!/bin/bash
# Round 4, trip 9: the exp2(polynomial) GELU epilogue (option gelu_poly): GPU tests over every bf16 pre-activation, the GEMM / step parity
# tests it touches, isolated fc1 GEMMs with the option off / on, interleaved step A/B with package power (also: N = 384 policy, serial step)
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
(timeout 600 python -m pytest tests/test_round4_gpu.py tests/test_kernels_gpu.py tests/test_round3_gpu.py -q -p no:cacheprovider -s -k "gelu or persistent or identical or epilogue" > $O/r4t9_tests_a.log 2>&1; echo "tests rc=$?" >> $O/r4t9_tests_a.log)
grep -E "passed|failed|FAILED|Error|rc=|gelu epilogue" $O/r4t9_tests_a.log | tail -12
(timeout 200 python tools/gemm_bench.py --reps 20 --cfgs 8.4 --no-wgrad --toggle gelu_poly --only fc1 > $O/r4t9_gelu_gemm.txt 2>&1; echo "rc=$?" >> $O/r4t9_gelu_gemm.txt)
cat $O/r4t9_gelu_gemm.txt | tail -6
(timeout 500 python tools/abab.py --power --arms "base;as:gelu_poly=0;n384:gemm_4w=2;serial:no_overlap=1" --rounds 6 --steps 6 --out $O/r4t9_abab.json > $O/r4t9_abab.md 2> $O/r4t9_abab.err; echo "rc=$?" >> $O/r4t9_abab.err)
cat $O/r4t9_abab.md; tail -2 $O/r4t9_abab.err
(timeout 900 python -m pytest tests/test_step_gpu.py tests/test_round2_gpu.py -q -p no:cacheprovider -x > $O/r4t9_tests_b.log 2>&1; echo "tests rc=$?" >> $O/r4t9_tests_b.log)
grep -E "passed|failed|FAILED|Error|rc=" $O/r4t9_tests_b.log | tail -8
