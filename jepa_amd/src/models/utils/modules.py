"""Transformer building blocks with the reference's module tree and state-dict names
(src/models/utils/modules.py:13-120): MLP{fc1,fc2}, Attention{qkv,proj}, Block{norm1,attn,norm2,mlp}.

The nn.Linear / nn.LayerNorm children are parameter containers (same names, shapes and default
initialisation as the reference); the arithmetic runs in the gfx950 kernels through jepa_amd.engine.
`forward` on a single Block/MLP/Attention runs the HIP chain for inference on bf16/fp32 GPU tensors.
"""
import torch
import torch.nn as nn


class MLP(nn.Module):
    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU, drop=0.):
        super().__init__()
        if drop != 0.:
            raise NotImplementedError("dropout is 0 in every V-JEPA config; the HIP path does not implement it")
        if act_layer is not nn.GELU:
            raise NotImplementedError("only exact-erf nn.GELU is fused into the fc1 epilogue")
        out_features = out_features or in_features
        hidden_features = hidden_features or in_features
        self.fc1 = nn.Linear(in_features, hidden_features)
        self.act = act_layer()
        self.fc2 = nn.Linear(hidden_features, out_features)


class Attention(nn.Module):
    def __init__(self, dim, num_heads=8, qkv_bias=False, qk_scale=None, attn_drop=0., proj_drop=0., use_sdpa=True):
        super().__init__()
        if attn_drop != 0. or proj_drop != 0.:
            raise NotImplementedError("dropout is 0 in every V-JEPA config; the HIP path does not implement it")
        if qk_scale is not None and abs(qk_scale - (dim // num_heads) ** -0.5) > 1e-12:
            raise NotImplementedError("custom qk_scale is ignored by the reference's SDPA branch (modules.py:66-69)")
        self.num_heads = num_heads
        self.scale = (dim // num_heads) ** -0.5
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.proj = nn.Linear(dim, dim)
        self.use_sdpa = use_sdpa


class Block(nn.Module):
    def __init__(self, dim, num_heads, mlp_ratio=4., qkv_bias=False, qk_scale=None, drop=0., attn_drop=0.,
                 act_layer=nn.GELU, norm_layer=nn.LayerNorm, grid_size=None, grid_depth=None):
        super().__init__()
        self.norm1 = norm_layer(dim)
        self.attn = Attention(dim, num_heads=num_heads, qkv_bias=qkv_bias, qk_scale=qk_scale, attn_drop=attn_drop,
                              proj_drop=drop)
        self.norm2 = norm_layer(dim)
        self.mlp = MLP(in_features=dim, hidden_features=int(dim * mlp_ratio), act_layer=act_layer, drop=drop)
        for n in (self.norm1, self.norm2):
            if abs(n.eps - 1e-6) > 1e-12:
                raise NotImplementedError("the HIP chain assumes LayerNorm eps=1e-6 (all reference ViT factories)")
