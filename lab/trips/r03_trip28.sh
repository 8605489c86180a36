#!/bin/bash
# trip 28: what bounds the attention forward?  Diagnostic builds of attn_fwd_pipe_kernel (results are WRONG by design):
# 1 = no LDS-DMA inside the loop, 2 = no barrier, 3 = neither, 4 = no v_exp, 8 = no P.V MFMAs
mkdir -p gpurun_out
for v in "" diag1 diag2 diag3 diag4 diag8; do
  if [ -z "$v" ]; then unset VJ_LIB_VARIANT; else export VJ_LIB_VARIANT=$v; fi
  echo "== variant '$v'" >> gpurun_out/r3t28_diag.log
  timeout 120 python tools/attn_bench.py --reps 20 --only-fwd --shapes "tgt prd" 2>&1 | grep -v amdgpu >> gpurun_out/r3t28_diag.log
done
