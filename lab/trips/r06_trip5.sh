#!/bin/bash
# Round 6, trip 5: SQ counters of the attention backward pairs (16x16x32 | 32x32x16 phase-by-phase | 32x32x16 pipelined) on the predictor shapes
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
A="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE"
B="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VALU_TRANS_F32 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM GRBM_GUI_ACTIVE"
C="SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAVES SQ_LEVEL_WAVES GRBM_GUI_ACTIVE"
CMD="python tools/attn_bench.py --reps 1 --shapes prd"
for v in m16 m32plain m32pipe; do
  case $v in m16) E="VJ_ATTN_MFMA=16";; m32plain) E="VJ_ATTN_MFMA=32 VJ_ATTN_BWD32_PLAIN=1";; m32pipe) E="VJ_ATTN_MFMA=32";; esac
  for p in A B C; do
    eval "PM=\$$p"
    env $E timeout 300 rocprofv3 --pmc $PM --output-format csv -d $O/pmc_r06_attn/${v}_$p -o x -- $CMD > $O/pmc_r06_attn.${v}_$p.log 2>&1
  done
  python tools/pmc_util_summary.py $O/pmc_r06_attn/${v}_util.md "attention backward $v: SQ counters (tools/attn_bench.py --reps 1 --shapes prd)" $O/pmc_r06_attn/${v}_A $O/pmc_r06_attn/${v}_B $O/pmc_r06_attn/${v}_C > /dev/null
  cat $O/pmc_r06_attn/${v}_util.md | head -30
done
find $O/pmc_r06_attn -name "*.csv" -size +8M -delete
