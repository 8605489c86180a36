"""Attentive probe on frozen V-JEPA features with the reference's classes, constructor arguments and state-dict names
(src/models/attentive_pooler.py:21-136; CrossAttention / CrossAttentionBlock from src/models/utils/modules.py:123-181), TRAINABLE:
forward and backward run on the gfx950 kernels behind the C ABI (bf16 MFMA GEMMs with fused bias / GELU / residual epilogues,
fp32-statistics LayerNorm, transpose-free weight gradients, the few-query cross-attention of csrc/xattn.hip) inside two autograd
nodes, so `loss.backward()` + any torch optimizer of the reference's eval loop (evals/video_classification_frozen/eval.py:298-352)
work unchanged.  Parameters stay fp32 nn.Parameters (the reference's layout); every step casts the five matrices to bf16.

What is supported is what the reference's evals instantiate: AttentiveClassifier(embed_dim, num_heads, depth=1, num_classes)
(eval.py:205-210) -- one query token, complete_block=True, no extra self-attention blocks; other settings raise.

Reference behaviours kept on purpose: CrossAttention owns a `proj` Linear that its forward never applies (modules.py:156-157), so
`xattn.proj.*` exist in the state dict, receive no gradient and do not influence the output; nn.LayerNorm default eps 1e-5.
"""
import math

import torch
import torch.nn as nn

from ...hip import ops
from ..utils.tensors import trunc_normal_
from .utils.modules import MLP


class CrossAttention(nn.Module):
    """Parameter container of modules.py:123-138 (q, kv, proj -- proj is never applied by the reference's forward)."""

    def __init__(self, dim, num_heads=12, qkv_bias=False, use_sdpa=True):
        super().__init__()
        self.num_heads = num_heads
        self.scale = (dim // num_heads) ** -0.5
        self.q = nn.Linear(dim, dim, bias=qkv_bias)
        self.kv = nn.Linear(dim, int(dim * 2), bias=qkv_bias)
        self.proj = nn.Linear(dim, dim)
        self.use_sdpa = use_sdpa


class CrossAttentionBlock(nn.Module):
    """Parameter container of modules.py:160-175 (norm1, xattn, norm2, mlp)."""

    def __init__(self, dim, num_heads, mlp_ratio=4., qkv_bias=False, act_layer=nn.GELU, norm_layer=nn.LayerNorm):
        super().__init__()
        self.norm1 = norm_layer(dim)
        self.xattn = CrossAttention(dim, num_heads=num_heads, qkv_bias=qkv_bias)
        self.norm2 = norm_layer(dim)
        self.mlp = MLP(in_features=dim, hidden_features=int(dim * mlp_ratio), act_layer=act_layer)


def _bf(t):
    return t.detach().to(torch.bfloat16).contiguous()


def _f32(t):
    return None if t is None else t.detach().float().contiguous()


def _wT(w_bf16):
    """[N_out, N_in] bf16 -> the dgrad operand W^T as a [N_in, N_out] view of the 64-padded transposed copy."""
    return ops.transpose(w_bf16)[:, :w_bf16.shape[0]]


def _zeros_like_param(p):
    return torch.zeros(p.shape, dtype=torch.float32, device=p.device)


class _PoolerFn(torch.autograd.Function):
    """AttentivePooler.forward (attentive_pooler.py:96-102) for one query token:  q0 -> q0 + xattn(q0, norm1(x)) -> + mlp(norm2(.))."""

    NAMES = ("query_tokens", "norm1.weight", "norm1.bias", "xattn.q.weight", "xattn.q.bias", "xattn.kv.weight", "xattn.kv.bias",
             "norm2.weight", "norm2.bias", "mlp.fc1.weight", "mlp.fc1.bias", "mlp.fc2.weight", "mlp.fc2.bias")

    @staticmethod
    def forward(ctx, x, heads, eps, qt, n1w, n1b, qw, qb, kvw, kvb, n2w, n2b, f1w, f1b, f2w, f2b):
        B, N, D = x.shape
        hd = D // heads
        x2 = x.detach().reshape(B * N, D).to(torch.bfloat16).contiguous()
        q0 = _bf(qt).reshape(1, D)
        wq, wkv, w1, w2 = _bf(qw), _bf(kvw), _bf(f1w), _bf(f2w)
        qh = ops.gemm_nt(q0, wq, bias=_f32(qb))                                                 # [1, D]: the same row for every sample
        xn, mean1, rstd1 = ops.layernorm_fwd(x2, _f32(n1w), _f32(n1b), eps)
        kv = ops.gemm_nt(xn, wkv, bias=_f32(kvb))                                              # packed [B, N, 2, H, hd]
        q1, lse = ops.xattn_fwd(qh, kv, B, 1, N, heads, hd, hd ** -0.5, resid=q0, shared_q=True)   # q0 + softmax(q k^T) v
        q1n, mean2, rstd2 = ops.layernorm_fwd(q1, _f32(n2w), _f32(n2b), eps)
        dgelu = torch.empty((B, w1.shape[0]), dtype=torch.bfloat16, device=x.device)
        g = ops.gemm_nt(q1n, w1, bias=_f32(f1b), aux_out=dgelu, epilogue=ops.EPI_GELU)
        q2 = ops.gemm_nt(g, w2, bias=_f32(f2b), residual=q1)
        ctx.saved = (x2, xn, mean1, rstd1, kv, qh, q0, lse, q1, q1n, mean2, rstd2, dgelu, g, wq, wkv, w1, w2)
        ctx.meta = (B, N, D, heads, hd, eps)
        ctx.params = (qt, n1w, n1b, qw, qb, kvw, kvb, n2w, n2b, f1w, f1b, f2w, f2b)
        return q2.float().view(B, 1, D)

    @staticmethod
    def backward(ctx, dout):
        x2, xn, mean1, rstd1, kv, qh, q0, lse, q1, q1n, mean2, rstd2, dgelu, g, wq, wkv, w1, w2 = ctx.saved
        B, N, D, heads, hd, eps = ctx.meta
        qt, n1w, n1b, qw, qb, kvw, kvb, n2w, n2b, f1w, f1b, f2w, f2b = ctx.params
        dev = x2.device
        with torch.no_grad():
            dq2 = dout.reshape(B, D).to(torch.bfloat16).contiguous()
            # mlp.fc2 (+ residual), mlp.fc1 (fused GELU backward), norm2
            g_f2w = ops.gemm_wgrad_tn(dq2, g, _zeros_like_param(f2w))
            g_f2b = ops.colsum(dq2, _zeros_like_param(f2b))
            du = ops.gemm_nt(dq2, _wT(w2), aux_in=dgelu, epilogue=ops.EPI_DGELU)
            g_f1w = ops.gemm_wgrad_tn(du, q1n, _zeros_like_param(f1w))
            g_f1b = ops.colsum(du, _zeros_like_param(f1b))
            dq1n = ops.gemm_nt(du, _wT(w1))
            g_n2w, g_n2b = _zeros_like_param(n2w), _zeros_like_param(n2b)
            dq1 = ops.layernorm_bwd(dq1n, q1, _f32(n2w), mean2, rstd2, g_n2w, g_n2b, dres=dq2)   # + the residual path of q1
            # q1 = q0 + y: the query token collects the batch sum; y goes back through the cross-attention
            g_qt = ops.colsum(dq1, torch.zeros(D, dtype=torch.float32, device=dev))
            dqh, dkv = ops.xattn_bwd(qh, kv, dq1, lse, B, N, heads, hd, hd ** -0.5, shared_q=True)
            g_qb = ops.colsum(dqh, torch.zeros(D, dtype=torch.float32, device=dev))              # the projected query is shared: batch sum
            dqh1 = torch.empty((1, D), dtype=torch.bfloat16, device=dev)
            ops.cast_bf16(g_qb, dqh1.view(-1))
            g_qw = ops.gemm_wgrad_tn(dqh1, q0, _zeros_like_param(qw))                              # outer product dqh^T q0
            ops.colsum(ops.gemm_nt(dqh1, _wT(wq)), g_qt, accumulate=True)                          # ... and through q = Linear(q0)
            # kv = Linear(norm1(x)): weight / bias gradients, then norm1's affine parameters (x itself is frozen)
            g_kvw = ops.gemm_wgrad_tn(dkv, xn, _zeros_like_param(kvw))
            g_kvb = ops.colsum(dkv, _zeros_like_param(kvb))
            dxn = ops.gemm_nt(dkv, _wT(wkv))
            g_n1w, g_n1b = _zeros_like_param(n1w), _zeros_like_param(n1b)
            ops.layernorm_bwd(dxn, x2, _f32(n1w), mean1, rstd1, g_n1w, g_n1b)
        if qb is None:
            g_qb = None
        if kvb is None:
            g_kvb = None
        return (None, None, None, g_qt.view_as(qt), g_n1w, g_n1b, g_qw, g_qb, g_kvw, g_kvb, g_n2w, g_n2b, g_f1w, g_f1b, g_f2w,
                g_f2b)


class _LinearFn(torch.autograd.Function):
    """nn.Linear on [B, D] rows (the classifier head, attentive_pooler.py:130-135); the class dimension is padded to a multiple of
    64 inside (zero rows / zero bias) so that any num_classes meets the GEMM's N % 4 and the dgrad's K % 32."""

    @staticmethod
    def forward(ctx, x, w, b):
        C, D = w.shape
        Cp = ops.pad64(C)
        xb = x.detach().reshape(-1, D).to(torch.bfloat16).contiguous()
        wp = torch.zeros((Cp, D), dtype=torch.bfloat16, device=x.device)
        wp[:C] = w.detach().to(torch.bfloat16)
        bp = torch.zeros(Cp, dtype=torch.float32, device=x.device)
        if b is not None:
            bp[:C] = b.detach().float()
        y = ops.gemm_nt(xb, wp, bias=bp)
        ctx.saved = (xb, wp)
        ctx.meta = (C, D, Cp, x.shape, b is not None)
        return y[:, :C].float().reshape(*x.shape[:-1], C)

    @staticmethod
    def backward(ctx, dy):
        xb, wp = ctx.saved
        C, D, Cp, xshape, has_b = ctx.meta
        with torch.no_grad():
            dl = torch.zeros((xb.shape[0], Cp), dtype=torch.bfloat16, device=xb.device)
            dl[:, :C] = dy.reshape(-1, C).to(torch.bfloat16)
            gw = ops.gemm_wgrad_tn(dl, xb, torch.zeros((Cp, D), dtype=torch.float32, device=xb.device))[:C]
            gb = ops.colsum(dl, torch.zeros(Cp, dtype=torch.float32, device=xb.device))[:C] if has_b else None
            dx = ops.gemm_nt(dl, _wT(wp)).float().reshape(xshape)
        return dx, gw.contiguous(), None if gb is None else gb.contiguous()


class AttentivePooler(nn.Module):
    """ Attentive Pooler (attentive_pooler.py:21-102) """

    def __init__(self, num_queries=1, embed_dim=768, num_heads=12, mlp_ratio=4.0, depth=1, norm_layer=nn.LayerNorm,
                 init_std=0.02, qkv_bias=True, complete_block=True):
        super().__init__()
        if num_queries != 1 or depth != 1 or not complete_block:
            raise NotImplementedError(
                "jepa_amd builds the probe the reference's evals instantiate: num_queries=1, depth=1, complete_block=True "
                "(evals/video_classification_frozen/eval.py:205-210)")
        if embed_dim % 32 != 0 or (embed_dim // num_heads) % 8 != 0 or embed_dim // num_heads > 128:
            raise NotImplementedError("embed_dim must be a multiple of 32 and head_dim a multiple of 8, at most 128")
        self.query_tokens = nn.Parameter(torch.zeros(1, num_queries, embed_dim))
        self.complete_block = complete_block
        self.cross_attention_block = CrossAttentionBlock(dim=embed_dim, num_heads=num_heads, mlp_ratio=mlp_ratio,
                                                         qkv_bias=qkv_bias, norm_layer=norm_layer)
        self.blocks = None
        self.init_std = init_std
        trunc_normal_(self.query_tokens, std=self.init_std)
        self.apply(self._init_weights)
        self._rescale_blocks()

    def _rescale_blocks(self):
        # attentive_pooler.py:68-81 with layer_id = 1: proj (never applied, but rescaled all the same) and fc2 / sqrt(2)
        self.cross_attention_block.xattn.proj.weight.data.div_(math.sqrt(2.0))
        self.cross_attention_block.mlp.fc2.weight.data.div_(math.sqrt(2.0))

    def _init_weights(self, m):
        if isinstance(m, nn.Linear):
            trunc_normal_(m.weight, std=self.init_std)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, nn.LayerNorm):
            nn.init.constant_(m.bias, 0)
            nn.init.constant_(m.weight, 1.0)

    def forward(self, x):
        """x: [B, N, D] frozen-encoder tokens (any float dtype, GPU) -> [B, 1, D] fp32."""
        if not x.is_cuda:
            raise ValueError("AttentivePooler: jepa_amd computes only on the GPU through libvjepa_hip.so (no CPU fallback)")
        if x.requires_grad and torch.is_grad_enabled():
            raise NotImplementedError("AttentivePooler: the features come from a FROZEN encoder (eval.py:330-337 runs it under "
                                      "torch.no_grad()); no gradient is propagated into them")
        blk = self.cross_attention_block
        eps = blk.norm1.eps
        return _PoolerFn.apply(x, blk.xattn.num_heads, eps, self.query_tokens, blk.norm1.weight, blk.norm1.bias, blk.xattn.q.weight,
                               blk.xattn.q.bias, blk.xattn.kv.weight, blk.xattn.kv.bias, blk.norm2.weight, blk.norm2.bias,
                               blk.mlp.fc1.weight, blk.mlp.fc1.bias, blk.mlp.fc2.weight, blk.mlp.fc2.bias)


class AttentiveClassifier(nn.Module):
    """ Attentive Classifier (attentive_pooler.py:105-136) """

    def __init__(self, embed_dim=768, num_heads=12, mlp_ratio=4.0, depth=1, norm_layer=nn.LayerNorm, init_std=0.02,
                 qkv_bias=True, num_classes=1000, complete_block=True):
        super().__init__()
        self.pooler = AttentivePooler(num_queries=1, embed_dim=embed_dim, num_heads=num_heads, mlp_ratio=mlp_ratio, depth=depth,
                                      norm_layer=norm_layer, init_std=init_std, qkv_bias=qkv_bias, complete_block=complete_block)
        self.linear = nn.Linear(embed_dim, num_classes, bias=True)

    def forward(self, x):
        x = self.pooler(x).squeeze(1)
        return _LinearFn.apply(x, self.linear.weight, self.linear.bias)
