"""Helpers shared by the GPU test files (imported, never collected)."""
import torch


def bf(x):
    return x.to(torch.bfloat16)


def sdpa_ref(qkv, B, S, H, hd):
    """fp32 F.scaled_dot_product_attention over a packed [B*S, 3*H*hd] qkv (reference Attention.forward, modules.py:61-78)."""
    q, k, v = qkv.float().view(B, S, 3, H, hd).permute(2, 0, 3, 1, 4)
    o = torch.nn.functional.scaled_dot_product_attention(q, k, v)
    return o.transpose(1, 2).reshape(B * S, H * hd)


ATTN_SHAPES = [(2, 20, 3, 64), (2, 64, 3, 32), (1, 107, 16, 64), (2, 366, 16, 64), (1, 1568, 4, 64),
               (2, 300, 16, 24), (1, 1113, 4, 24), (1, 200, 2, 80), (3, 52, 3, 32), (1, 129, 2, 128),
               # ViT-H head_dim 80 on the native 96-wide class (3 k-steps, 5 output tiles): ragged, multi-tile, and the
               # full 384^2 x 16 frame sequence of BASELINE configs[4] (8 x 24 x 24 = 4608 tokens, 16 heads)
               (2, 63, 16, 80), (1, 1568, 2, 80), (1, 4608, 16, 80), (2, 577, 3, 72),
               # head_dim 24 / 32 on the re-swizzled 64-byte-row images: tile boundaries +-1
               (1, 64, 2, 24), (1, 65, 2, 24), (2, 127, 2, 32), (1, 1208, 16, 24)]


class opt:
    """with opt("name", value): ... restores the previous value of a run-time option (vj_set_option)."""

    def __init__(self, name, value):
        self.name, self.value = name, value

    def __enter__(self):
        from jepa_amd.hip.lib import set_option
        self.old = set_option(self.name, self.value)

    def __exit__(self, *a):
        from jepa_amd.hip.lib import set_option
        set_option(self.name, self.old)


def fused_bias_mask(tr, arena=None):
    """bool mask over a parameter arena (default: the trainable one): True on the qkv / fc1 biases (the only gradients option
    bias_fuse changes)."""
    arena = tr.arena if arena is None else arena
    lo = getattr(arena, "lo", 0)                 # the EMA target arena covers the encoder range [lo, hi) of the trainer arena
    m = torch.zeros(arena.P.numel(), dtype=torch.bool, device=arena.P.device)
    for name, sl in arena.slots.items():
        if name.endswith("attn.qkv.bias") or name.endswith("mlp.fc1.bias"):
            m[sl.off - lo:sl.off - lo + sl.numel] = True
    return m
