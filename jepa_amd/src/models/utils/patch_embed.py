"""Tubelet patch embedding with the reference's parameter names (src/models/utils/patch_embed.py:31-57).

`proj` is an nn.Conv3d kept as the parameter container ([D,3,tub,p,p] weight, [D] bias, reference initialisation and
state-dict names).  A Conv3d whose stride equals its kernel is a GEMM over non-overlapping tubelets: `vj_tubelet_pack`
reads the fp32 clip once and emits the bf16 [B*N, 3*tub*p*p] operand (element order = Conv3d weight order), the MFMA
GEMM adds the bias.  Inside VisionTransformer the pack also fuses the context-mask gather; the stand-alone `forward`
below is the unmasked, no-grad form."""
import torch
import torch.nn as nn

from ....hip import ops


class PatchEmbed3D(nn.Module):
    def __init__(self, patch_size=16, tubelet_size=2, in_chans=3, embed_dim=768):
        super().__init__()
        self.patch_size = patch_size
        self.tubelet_size = tubelet_size
        self.proj = nn.Conv3d(in_channels=in_chans, out_channels=embed_dim,
                              kernel_size=(tubelet_size, patch_size, patch_size),
                              stride=(tubelet_size, patch_size, patch_size))

    def forward(self, x, **kwargs):
        """x fp32 [B,C,T,H,W] on the GPU -> [B, N, D] tokens in (t', h', w') order
        (= proj(x).flatten(2).transpose(1, 2), patch_embed.py:54-57); inference only."""
        if not x.is_cuda:
            raise ValueError("PatchEmbed3D.forward: jepa_amd computes only on the GPU (no CPU fallback)")
        if torch.is_grad_enabled() and (x.requires_grad or self.proj.weight.requires_grad):
            raise NotImplementedError("PatchEmbed3D.forward is inference-only; train through VisionTransformer")
        B = x.shape[0]
        D = self.proj.weight.shape[0]
        tok = ops.tubelet_pack(x.float().contiguous(), self.tubelet_size, self.patch_size)
        w = self.proj.weight.detach().reshape(D, -1).to(torch.bfloat16).contiguous()
        y = ops.gemm_nt(tok, w, bias=self.proj.bias.detach().float().contiguous())
        return y.view(B, -1, D).to(x.dtype)
