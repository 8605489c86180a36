"""Transformer building blocks with the reference's module tree and state-dict names
(src/models/utils/modules.py:13-120): MLP{fc1,fc2}, Attention{qkv,proj}, Block{norm1,attn,norm2,mlp}.

The nn.Linear / nn.LayerNorm children are parameter containers (same names, shapes and default
initialisation as the reference).  Training never calls these modules one by one: the whole trunk runs as one
launch chain (jepa_amd.engine).  Their stand-alone `forward` is a no-grad inference path over the same gfx950
kernels (bf16 MFMA GEMMs with fused bias / GELU / residual epilogues, flash attention, fp32-statistics LayerNorm)
for code that composes blocks by hand, e.g. probes on frozen features; it raises if a gradient would be needed.
"""
import torch
import torch.nn as nn

from ....hip import ops


def _no_grad_only(module, x, what):
    if not x.is_cuda:
        raise ValueError(f"{what}: jepa_amd computes only on the GPU through libvjepa_hip.so (no CPU fallback)")
    if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in module.parameters())):
        raise NotImplementedError(
            f"{what}: the stand-alone module forward is inference-only; call it under torch.no_grad(), or train "
            "through VisionTransformer / VisionTransformerPredictor (one autograd node per trunk) or the fused Trainer")


def _bf16_weights(module):
    """bf16 copies of the Linear weights of `module`, refreshed when a parameter's version counter moved or the fused
    Trainer rewrote parameters through its arenas (raw-pointer writes that torch's version counters do not see)."""
    from ....engine.weights import ARENA_GENERATION, wait_pending_update
    wait_pending_update()   # a Trainer's deferred update must have run before its parameters are read
    key = (ARENA_GENERATION[0],) + tuple((p.data_ptr(), p._version) for p in module.parameters())
    cache = module.__dict__.get("_hip_w")
    if cache is None or cache[0] != key:
        cache = (key, {n: p.detach().to(torch.bfloat16).contiguous() for n, p in module.named_parameters()
                       if p.dim() == 2})
        module.__dict__["_hip_w"] = cache
    return cache[1]


def _f32(p):
    return None if p is None else p.detach().float().contiguous()


def _rows(x):
    return x.reshape(-1, x.shape[-1]).to(torch.bfloat16).contiguous()


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

    def _run(self, x2, w, residual=None):
        g = ops.gemm_nt(x2, w["fc1.weight"], bias=_f32(self.fc1.bias), epilogue=ops.EPI_GELU)   # fc1 + GELU fused
        return ops.gemm_nt(g, w["fc2.weight"], bias=_f32(self.fc2.bias), residual=residual)

    def forward(self, x):
        """fc2(GELU(fc1(x))) (modules.py:30-36); any float dtype in, same dtype out."""
        _no_grad_only(self, x, "MLP.forward")
        y = self._run(_rows(x), _bf16_weights(self))
        return y.view(*x.shape[:-1], y.shape[-1]).to(x.dtype)


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

    def _run(self, x2, B, N, w, residual=None):
        C = x2.shape[1]
        qkv = ops.gemm_nt(x2, w["qkv.weight"], bias=_f32(self.qkv.bias))     # packed [B,N,3,H,hd], read strided
        o, _ = ops.attn_fwd(qkv, B, N, self.num_heads, C // self.num_heads, self.scale, save_lse=False)
        return ops.gemm_nt(o, w["proj.weight"], bias=_f32(self.proj.bias), residual=residual)

    def forward(self, x, mask=None):
        """proj(SDPA(split(qkv(x)))) (modules.py:61-78); `mask` is accepted and ignored exactly like the reference."""
        _no_grad_only(self, x, "Attention.forward")
        B, N, C = x.shape
        y = self._run(_rows(x), B, N, _bf16_weights(self))
        return y.view(B, N, C).to(x.dtype)


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

    def forward(self, x, mask=None):
        """x + attn(norm1(x)), then + mlp(norm2(.)) (modules.py:114-120); residual stream in bf16 like the trunk chain,
        both residual adds fused into the proj / fc2 GEMM epilogues."""
        _no_grad_only(self, x, "Block.forward")
        B, N, C = x.shape
        x2 = _rows(x)
        y1, _, _ = ops.layernorm_fwd(x2, _f32(self.norm1.weight), _f32(self.norm1.bias), self.norm1.eps, save_stats=False)
        x1 = self.attn._run(y1, B, N, _bf16_weights(self.attn), residual=x2)
        y2, _, _ = ops.layernorm_fwd(x1, _f32(self.norm2.weight), _f32(self.norm2.bias), self.norm2.eps, save_stats=False)
        out = self.mlp._run(y2, _bf16_weights(self.mlp), residual=x1)
        return out.view(B, N, C).to(x.dtype)
