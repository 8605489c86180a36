// Run-time tuning switches of the library (vj_set_option / vj_get_option, include/vjepa_hip.h).
//
// Every switch selects between kernels that compute the SAME result (bit-identical unless the table below says
// otherwise); they exist so that A/B measurements can be interleaved inside ONE process on one GPU (tools/abab.py):
// box-to-box and thermal drift on MI355X is larger than most kernel-level deltas.  Initial values come from the
// environment variable of the same name in upper case with a VJ_ prefix (VJ_GEMM_4W=1 ...), read once.
// Every switch has an accepted value range (options.cpp): vj_set_option rejects anything else.  Switches are process-global and
// read at ENQUEUE time: change them only BETWEEN optimisation steps (changing e.g. wgrad_group between the micro-batches of
// one step would change the fp32 summation order mid-accumulation).
#pragma once

enum VjOpt {
  VJ_OPT_GEMM_FWD_FLAGS = 0,   // vj_gemm_bf16_nt flags for the chains' forward GEMMs (0 = automatic selection)
  VJ_OPT_GEMM_DGRAD_FLAGS,     // ... for the chains' dgrad GEMMs
  VJ_OPT_GEMM_4W,              // 1: every forward / dgrad GEMM on the 4-wave 256x128 kernel; 2: only N = 384 outputs (a 256-wide
                               // tile wastes a third there; -0.10 ms/step, 6 of 6 rounds: profiles/r03_abab_n384_policy.md);
                               // 3 / 4 / 5 (round 5): the PERSISTENT two-workgroups-per-CU form (gemm_nt_4wp_kernel) for every shape the
                               // persistent 8-phase kernel takes / only K <= 512 / only K <= 1024 and N <= 1152
  VJ_OPT_GEMM_PERSIST,         // 1 (default): persistent 8-phase kernel (gemm8p.hip) where it applies (single-round shapes
                               // included), trimmed grid; 2: one
                               // workgroup per CU; 0: always one tile per workgroup (gemm8.hip)
  VJ_OPT_WGRAD_TN,             // 1 (default): transpose-free weight gradients (gemm8_tn.hip); 0: transposes + NT GEMM
  VJ_OPT_WGRAD_GROUP,          // 1 (default): the four weight gradients of a block in ONE launch (vj_gemm_bf16_tn_grouped);
                               // 0: one launch each (different fp32 summation order: results agree to rounding, not bitwise)
  VJ_OPT_WGRAD_SLOW_ISSUE,     // 1: the TN kernel's K loop issues its parts through the generic address path (A/B only)
  VJ_OPT_ATTN_DKDV_KT,         // 16-key tiles per wave in the attention dK/dV kernel: 0 (default) per head-dim class, 1 / 2 forced, 4 (round 5) at head_dim <= 32
  VJ_OPT_GEMM_DBG,             // diagnostics of the GEMM kernels (bit0 drop the epilogue, bit1 unstaged stores, bit2 phase stamps of the persistent kernel: tools/gemm_stamps.py)
  VJ_OPT_ATTN_SOFTMAX,         // 1 (A/B only): attention kernels with the soft-max scale folded into the stationary operand
                               // and the score accumulators seeded with -max / -lse (no per-score FMA, no per-tile row maximum).  With a
                               // positive scale the forward / dQ kernels fold c into Q and dK/dV folds it into K: the backward's scores
                               // then differ from the ones lse2 was built from by one bf16 rounding of the operand (2^-9 |s|), so mode 1
                               // is for measurements, not for training;
                               // 0: the round-3 kernels (results agree to bf16 rounding, not bitwise);
                               // 2 (DEFAULT since round 4): as 1, but inside the block chains the scale reaches q in the qkv GEMM's epilogue (vj_gemm_bf16_nt
                               // epilogue 4: one rounding of c*q, no second rounding of the stationary operand) and the attention
                               // entry points are told so through a NEGATIVE scale argument
  VJ_OPT_BIAS_FUSE,            // 1 (default, round 4): qkv / fc1 bias gradients from column partials written by the kernels that
                               // PRODUCE dY (attention backward, fc2-dgrad epilogue), one reduction launch per block;
                               // 0: stand-alone column-sum kernels re-reading dY
  VJ_OPT_GELU_POLY,            // 1 (default, round 4): the GELU epilogues take Phi(-|x|) as exp2 of a degree-6 polynomial in min(|x|, 5)
                               // (6 FMAs + the one v_exp; no v_rcp, three multiplies fewer; closer to the correctly rounded bf16 erf-GELU
                               // than 0:) Abramowitz-Stegun 7.1.26.  Results agree to one bf16 ulp on < 0.2 % of the inputs, not bitwise
  VJ_OPT_GEMM_SCHED,           // (values 4 and 8 only) load / compute section pairs per K-tile of the persistent NT GEMM: 8 = four pairs of 16 MFMAs (round 3),
                               // 4 = two pairs of 32 MFMAs (round 4: half the section boundaries); bit-identical results
  VJ_OPT_ATTN_PSUM,            // 1 (default): the forward takes its soft-max row sums from the matrix pipe (head_dim 24: the V pad column of
                               // the P.V MFMA; other head sizes: an all-ones operand, two extra MFMAs per key tile); 0: vector adds
  VJ_OPT_ATTN_MERGE,           // 1 (default): the chains launch attention ONCE per block for all segments (masks) of the batch
                               // (vj_attn_fwd_segs / vj_attn_bwd_segs); 0: one launch (pair) per segment.  Bit-identical results
  VJ_OPT_LN_BWD_PREFETCH,      // 1 (default, round 4): the LayerNorm backward requests x | dy | dres | mean | rstd of its next row before it
                               // computes the current one; 0: when the row is needed.  Bit-identical results
  VJ_OPT_GEMM_RASTER,          // tile order of the persistent NT GEMM (gemm_common.hpp tile_of_raster): bits 0-7 group size (0 = 8), bit 8 = groups of
                               // COLUMN tiles walking down the rows instead of groups of row tiles sweeping the columns; 511 = column groups of six for
                               // K >= 1024, the row-grouped order otherwise.  Default 260 (late round 4): column groups of FOUR -- an XCD's 32 concurrent tiles
                               // are still 8 rows x 4 columns, but its band walks down the rows of one column group, so the B panels stay in its L2 and
                               // every A panel streams through once per column group: fabric reads of the encoder shapes -14 ... -38 %, step -0.6 ... -0.8 ms
                               // (profiles/r04_gemm_raster.md).  0 = the order of rounds 2-4 (groups of 8 row tiles).  Bit-identical results
  VJ_OPT_ATTN_DQ_QW,           // 16-query tiles per wave in the attention dQ kernel: 0 (default) = 2 (128 queries per workgroup); 4 = four at head_dim <= 32
                               // (256 queries per workgroup: half the LDS instructions per MFMA).  dqkv bit-identical; the dQ column partials regroup
  VJ_OPT_GEMM_NT,              // non-temporal hint on the persistent NT GEMM's LDS-DMA: 1 = on the operand that only streams through an XCD's L2 under
                               // the current tile order (A for column groups), 2 = on the other one (control), 0 = none.  Bit-identical results
  VJ_OPT_GEMM_DYN,             // 1: the persistent NT GEMM hands out every tile beyond a workgroup's first two from per-XCD atomic counters (gemm8p.hip);
                               // 0: static round-robin lists.  Bit-identical results
  VJ_OPT_ADAM_GRID,            // cap on the workgroup count of the guarded fused AdamW / EMA kernel (0 = none: 8 workgroups per CU); A/B of the
                               // range-wise update running beside the next step's forward (Trainer(overlap_update)).  Same results
  VJ_OPT_WS_GUARD,             // diagnostics: 1 = 256-byte guard gaps behind every member of the chain workspaces, poisoned by the chain calls and
                               // inspected by vj_ws_guard_check (tests/test_round5_gpu.py).  Changes the workspace sizes: set it before the first step
  VJ_OPT_GEMM_EPI_PRE,         // persistent NT GEMM, form of the epilogue (gemm_common.hpp gemm_epilogue_staged PRE; all bit-identical): 0 = straight passes, a row
                               // operand (residual, saved gelu') requested one 16-row block ahead; 1 = all eight blocks before the epilogue's single vmcnt(0)
                               // (MFMA layout); 2 = as sixteen full-line 16-byte loads re-laid-out through the staging area; 3 = 2 without the blanket
                               // wait; 4 (default) = 3 + software-pipelined passes, scalar row pointers, no-bias variants, for EVERY epilogue;
                               // 5 / 6 = diagnostic copies of 4 inside the phase-stamping kernel only (no stores / no LDS round trip: wrong outputs)
  VJ_OPT_COUNT
};

int vj_opt(int id);   // current value (relaxed atomic load; safe from any thread)
