#!/bin/bash
# SQ counter passes (MFMA utilisation etc.) over the attention and GEMM micro-benchmarks.  Counters only: no tracing
# domains in the same run (see the gpurun rules).  Usage on the GPU box: bash tools/pmc_mfma.sh gpurun_out/pmc_r02
set -u
OUT=${1:-gpurun_out/pmc_r02}
export TMPDIR=/tmp
A="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE"
B="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VALU_TRANS_F32 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM GRBM_GUI_ACTIVE"
for what in attn gemm; do
  if [ $what = attn ]; then CMD="python tools/attn_bench.py --reps 1"; else CMD="python tools/gemm_bench.py --reps 1 --cfgs 8.4"; fi
  timeout 300 rocprofv3 --pmc $A --output-format csv -d $OUT/${what}_a -o a -- $CMD > $OUT.${what}_a.log 2>&1
  timeout 300 rocprofv3 --pmc $B --output-format csv -d $OUT/${what}_b -o b -- $CMD > $OUT.${what}_b.log 2>&1
done
python tools/pmc_util_summary.py $OUT/attn_util.md "Attention kernels: SQ counters (tools/attn_bench.py --reps 1)" $OUT/attn_a $OUT/attn_b > /dev/null
python tools/pmc_util_summary.py $OUT/gemm_util.md "GEMM kernels: SQ counters (tools/gemm_bench.py --reps 1 --cfgs 8.4: default selection, persistent kernel, 4-section schedule)" $OUT/gemm_a $OUT/gemm_b > /dev/null
