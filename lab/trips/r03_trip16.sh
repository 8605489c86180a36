#!/bin/bash
# trip 16: SQ counters of the TN weight-gradient kernel (single + no-split "grouped-equivalent" launches)
set -u
OUT=gpurun_out/pmc_r03
mkdir -p $OUT
export TMPDIR=/tmp
A="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE"
B="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VALU_TRANS_F32 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM GRBM_GUI_ACTIVE"
CMD="python tools/wgrad_group_probe.py"
timeout 300 rocprofv3 --pmc $A --output-format csv -d $OUT/tn_a -o a -- $CMD > $OUT.tn_a.log 2>&1
timeout 300 rocprofv3 --pmc $B --output-format csv -d $OUT/tn_b -o b -- $CMD > $OUT.tn_b.log 2>&1
python tools/pmc_util_summary.py $OUT/tn_util.md "TN weight-gradient kernel: SQ counters (tools/wgrad_group_probe.py)" $OUT/tn_a $OUT/tn_b > /dev/null
