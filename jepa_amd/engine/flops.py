"""Algorithmic FLOP model of one V-JEPA pretraining step (matmul FLOPs only, 2*M*N*K; backward = 2x forward;
softmax / LayerNorm / GELU / optimizer FLOPs excluded) -- the figure the MFMA roofline fraction is quoted on.

  F_blk(s, d) = 24*s*d^2 + 4*s^2*d            one transformer block on a sequence of s tokens (mlp_ratio 4)
  F_tgt       = depth*F_blk(N, D) + 2*N*Kpe*D                            target encoder, forward only
  F_ctx       = sum_i depth*F_blk(Ke_i, D) + 2*Ke_i*Kpe*D                context encoder (kept tokens only)
  F_pred      = sum_i pdepth*F_blk(Ke_i+Kp_i, Dp) + 2*Ke_i*D*Dp + 2*Kp_i*Dp*D
  F_step      = B * (F_tgt + 3*(F_ctx + F_pred))
"""


def block_flops(s, d):
    return 24 * s * d * d + 4 * s * s * d


def step_flops(embed_dim, depth, pred_dim, pred_depth, num_patches, patch_k, B, Ke_list, Kp_list):
    f_tgt = depth * block_flops(num_patches, embed_dim) + 2 * num_patches * patch_k * embed_dim
    f_ctx = sum(depth * block_flops(ke, embed_dim) + 2 * ke * patch_k * embed_dim for ke in Ke_list)
    f_pred = sum(pred_depth * block_flops(ke + kp, pred_dim) + 2 * ke * embed_dim * pred_dim
                 + 2 * kp * pred_dim * embed_dim for ke, kp in zip(Ke_list, Kp_list))
    return B * (f_tgt + 3 * (f_ctx + f_pred))
