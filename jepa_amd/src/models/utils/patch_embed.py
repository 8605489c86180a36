"""Tubelet patch embedding with the reference's parameter names (src/models/utils/patch_embed.py:31-57).

`proj` is an nn.Conv3d used purely as the parameter container ([D,3,tub,p,p] weight, [D] bias, reference
initialisation); the non-overlapping convolution itself runs as vj_tubelet_pack + the bf16 MFMA GEMM."""
import torch.nn as nn


class PatchEmbed3D(nn.Module):
    def __init__(self, patch_size=16, tubelet_size=2, in_chans=3, embed_dim=768):
        super().__init__()
        self.patch_size = patch_size
        self.tubelet_size = tubelet_size
        self.proj = nn.Conv3d(in_channels=in_chans, out_channels=embed_dim,
                              kernel_size=(tubelet_size, patch_size, patch_size),
                              stride=(tubelet_size, patch_size, patch_size))
