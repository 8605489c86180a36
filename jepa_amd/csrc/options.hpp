// Run-time switches of the library (vj_set_option / vj_get_option, include/vjepa_hip.h).
//
// Twelve of them since round 6: two kernel-selection flag words, one A/B control per kernel family (the selected forms compute the SAME
// result -- bit-identical unless the table says otherwise), and two diagnostics.  They exist so that A/B measurements can be interleaved
// inside ONE process on one GPU (tools/abab.py): box-to-box and thermal drift on MI355X is larger than most kernel-level deltas.  Initial
// values come from the environment variable of the same name in upper case with a VJ_ prefix (VJ_GEMM_4W=1 ...), read once.  Every switch
// has an accepted value set (options.cpp): vj_set_option rejects anything else.  Switches are process-global and read at ENQUEUE time:
// change them only BETWEEN optimisation steps.
// Rounds 3-5 carried 23 switches; the measured-negative forms behind the other eleven (and their bit-identity tests) were removed in
// round 6 -- what they were and what they measured is in docs/history/ and under lab/patches/.
#pragma once

enum VjOpt {
  VJ_OPT_GEMM_FWD_FLAGS = 0,   // vj_gemm_bf16_nt flags for the chains' forward GEMMs (0 = automatic selection; gemm.hip dispatch_gemm)
  VJ_OPT_GEMM_DGRAD_FLAGS,     // ... for the chains' dgrad GEMMs
  VJ_OPT_GEMM_4W,              // 1: every forward / dgrad GEMM on the 4-wave 256x128 kernel with two workgroups per CU (gemm4w.hip: faster on a single
                               // stream -- frozen-encoder inference --, 1.2 - 5.4 % slower in the two-stream training step); 0 (default): automatic
  VJ_OPT_GEMM_PERSIST,         // persistent 256x256 kernel (gemm8p.hip) where it applies: 2 (default since late round 6): one workgroup per CU; 1: the smallest
                               // grid that keeps the number of tile rounds (rounds 3 - 6: left CUs to the other stream; -0.2 ms for 2 now, 9 of 10 rounds);
                               // 3: as 1, but N % 256 == 128 launches recompute the overlap instead of running half tiles (A/B control of round 6);
                               // 0: always one tile per workgroup (gemm8.hip) -- the bit-identity control of the persistent kernel
  VJ_OPT_WGRAD_TN,             // 1 (default): transpose-free weight gradients (gemm8_tn.hip); 0: transposes + NT split-K GEMM (the cross-check route)
  VJ_OPT_WGRAD_GROUP,          // 1 (default): the four weight gradients of a block in ONE launch (vj_gemm_bf16_tn_grouped);
                               // 0: one launch each (different fp32 summation order: results agree to rounding, not bitwise)
  VJ_OPT_GEMM_DBG,             // diagnostics of the GEMM kernels (bit0 drop the epilogue, bit1 unstaged stores, bit2 phase stamps: tools/gemm_stamps.py)
  VJ_OPT_ATTN_SOFTMAX,         // where the soft-max scale scale * log2(e) enters (values 1 and 2): 2 (default) = inside the block chains the qkv GEMM
                               // multiplies its q columns by it before their ONE bf16 rounding (vj_gemm_bf16_nt epilogue 4) and the attention entry
                               // points are told so through a NEGATIVE scale argument; 1 = the attention kernels fold it into their stationary
                               // operand themselves (one more bf16 rounding of q / k: what a stand-alone call with a positive scale gets anyway)
  VJ_OPT_BIAS_FUSE,            // 1 (default): qkv / fc1 bias gradients from column partials written by the kernels that PRODUCE dY (attention backward,
                               // fc2-dgrad epilogue), one reduction launch per block; 0: stand-alone column-sum kernels re-reading dY
  VJ_OPT_GEMM_RASTER,          // tile order of the persistent NT GEMM (gemm_common.hpp tile_of_raster): bits 0-7 group size (0 = 8), bit 8 = groups of
                               // COLUMN tiles walking down the rows instead of groups of row tiles sweeping the columns; 511 = column groups of six for
                               // K >= 1024, the row-grouped order otherwise.  Default 260: column groups of four (profiles/r04_gemm_raster.md).  Bit-identical
  VJ_OPT_WS_GUARD,             // diagnostics: 1 = 256-byte guard gaps behind every member of the chain workspaces, poisoned by the chain calls and
                               // inspected by vj_ws_guard_check.  Changes the workspace sizes: set it before the first step
  VJ_OPT_GEMM_EPI_PRE,         // epilogue form of the persistent NT GEMM (values 0 and 4, bit-identical): 4 (default) = row operand as full-line loads
                               // through the staging area + software-pipelined passes; 0 = straight passes (the A/B control: profiles/r05_epi_pipeline.md)
  VJ_OPT_COUNT
};

int vj_opt(int id);   // current value (relaxed atomic load; safe from any thread)
