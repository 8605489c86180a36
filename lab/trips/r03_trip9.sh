#!/bin/bash
# Round 3, trip 9: LayerNorm forward with gamma/beta in LDS (tests + micro-bench), 1-rank reducer bit-identity, MFMA-utilisation
# counters of the persistent GEMM, in-step A/B is not possible for the LN change (no switch) -> bench line only.
export TMPDIR=/tmp
O=gpurun_out
(timeout 600 python -m pytest tests/test_round3_gpu.py tests/test_kernels_gpu.py -q -x -p no:cacheprovider -k "reducer_at_one_rank or layernorm or target_rows" > $O/r3t9_tests.log 2>&1; echo "tests rc=$?" >> $O/r3t9_tests.log)
grep -E "passed|failed|FAILED|ERROR|rc=|Error" $O/r3t9_tests.log | tail -8
(timeout 120 python tools/ln_bench.py > $O/r3t9_ln_bench.log 2>&1); cat $O/r3t9_ln_bench.log | tail -12
(timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-roofline-pass > $O/r3t9_bench.json 2> $O/r3t9_bench.err); grep timed $O/r3t9_bench.err | cut -c1-200
mkdir -p $O/pmc_r03
A="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE"
B="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VALU_TRANS_F32 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM GRBM_GUI_ACTIVE"
CMD="python tools/gemm_bench.py --reps 1 --cfgs 8.0"
timeout 300 rocprofv3 --pmc $A --output-format csv -d $O/pmc_r03/gemmp_a -o a -- $CMD > $O/pmc_r03.gemmp_a.log 2>&1
timeout 300 rocprofv3 --pmc $B --output-format csv -d $O/pmc_r03/gemmp_b -o b -- $CMD > $O/pmc_r03.gemmp_b.log 2>&1
python tools/pmc_util_summary.py $O/pmc_r03/gemmp_util.md "GEMM kernels, default selection with the persistent kernel: SQ counters (tools/gemm_bench.py --reps 1 --cfgs 8.0)" $O/pmc_r03/gemmp_a $O/pmc_r03/gemmp_b > /dev/null
head -12 $O/pmc_r03/gemmp_util.md
find $O/pmc_r03 -name "*.csv" -size +8M -delete
