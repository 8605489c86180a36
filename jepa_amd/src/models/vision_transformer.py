"""Video Vision Transformer with the reference's constructor, factories, attributes and state-dict names
(src/models/vision_transformer.py:21-307), computing on MI355X through the jepa_amd HIP kernels.

    vit = vit_large(img_size=224, patch_size=16, num_frames=16, tubelet_size=2, uniform_power=True)
    out = vit(clips)                  # [B, N, D]
    out = vit(clips, masks=[idx])     # [B*len(masks), K, D]   (reference contract: masks share K)

`forward` is differentiable (one autograd node per call whose backward is the hand-written layer chain), but
the pretraining step (jepa_amd.engine.step.Trainer) bypasses autograd entirely and shares this module's
parameters through flat arenas.
"""
import math
from functools import partial

import torch
import torch.nn as nn

from ...engine import hipmodule
from ...engine.layers import encoder_backward, encoder_forward
from ..utils.tensors import trunc_normal_
from .utils.modules import Block
from .utils.patch_embed import PatchEmbed3D
from .utils.pos_embs import get_3d_sincos_pos_embed


class VisionTransformer(nn.Module, hipmodule.HipModule):
    """ Vision Transformer (video) """

    def __init__(self, img_size=224, patch_size=16, num_frames=1, tubelet_size=2, in_chans=3, embed_dim=768,
                 depth=12, num_heads=12, mlp_ratio=4.0, qkv_bias=True, qk_scale=None, drop_rate=0.0,
                 attn_drop_rate=0.0, norm_layer=nn.LayerNorm, init_std=0.02, out_layers=None, uniform_power=False,
                 **kwargs):
        super().__init__()
        self.num_features = self.embed_dim = embed_dim
        self.num_heads = num_heads
        self.out_layers = out_layers
        self.input_size = img_size
        self.patch_size = patch_size
        self.num_frames = num_frames
        self.tubelet_size = tubelet_size
        self.is_video = num_frames > 1
        if not self.is_video:
            raise NotImplementedError("image (num_frames=1) ViTs belong to the frozen-eval path, outside the "
                                      "V-JEPA pretraining step this package accelerates")
        if in_chans != 3 or not qkv_bias or out_layers is not None:
            raise NotImplementedError("only in_chans=3, qkv_bias=True, out_layers=None (the pretraining setup)")
        grid_size = img_size // patch_size
        grid_depth = num_frames // tubelet_size
        self.patch_embed = PatchEmbed3D(patch_size=patch_size, tubelet_size=tubelet_size, in_chans=in_chans,
                                        embed_dim=embed_dim)
        self.num_patches = grid_depth * grid_size * grid_size
        self.uniform_power = uniform_power
        self.pos_embed = nn.Parameter(torch.zeros(1, self.num_patches, embed_dim), requires_grad=False)
        self.blocks = nn.ModuleList([
            Block(dim=embed_dim, num_heads=num_heads, mlp_ratio=mlp_ratio, qkv_bias=qkv_bias, qk_scale=qk_scale,
                  drop=drop_rate, act_layer=nn.GELU, grid_size=grid_size, grid_depth=grid_depth,
                  attn_drop=attn_drop_rate, norm_layer=norm_layer) for _ in range(depth)])
        self.norm = norm_layer(embed_dim)
        # ---- weights: sincos table, trunc-normal(0.02) matrices, zero biases, unit norms, depth rescale
        sincos = get_3d_sincos_pos_embed(embed_dim, grid_size, grid_depth, cls_token=False,
                                         uniform_power=uniform_power)
        self.pos_embed.data.copy_(torch.from_numpy(sincos).float().unsqueeze(0))
        self.init_std = init_std
        self.apply(self._init_weights)
        self._rescale_blocks()

    def _init_weights(self, m):
        if isinstance(m, (nn.Linear, nn.Conv2d, nn.Conv3d)):
            trunc_normal_(m.weight, std=self.init_std)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, nn.LayerNorm):
            nn.init.constant_(m.bias, 0)
            nn.init.constant_(m.weight, 1.0)

    def _rescale_blocks(self):
        for layer_id, layer in enumerate(self.blocks):
            layer.attn.proj.weight.data.div_(math.sqrt(2.0 * (layer_id + 1)))
            layer.mlp.fc2.weight.data.div_(math.sqrt(2.0 * (layer_id + 1)))

    def get_num_layers(self):
        return len(self.blocks)

    def no_weight_decay(self):
        return {}

    def interpolate_pos_encoding(self, x, pos_embed):
        _, _, T, H, W = x.shape
        if H == self.input_size and W == self.input_size and T == self.num_frames:
            return pos_embed
        raise NotImplementedError("pos-embed interpolation (non-native resolution) is an eval-time feature; the "
                                  "pretraining step always runs at the native clip size")

    # ---- compute ------------------------------------------------------------------------------------------
    def _hip_views(self, train):
        from ...engine.weights import encoder_views
        arena, prefix = self._hip_arena(train)
        cache = self.__dict__.setdefault("_hip_view_cache", {})
        key = (id(arena), prefix, bool(train), len(arena.wT))
        v = cache.get(key)
        if v is None:   # the views (and the C descriptor array hanging off them) only change with the arena
            cache.clear()
            v = cache[key] = encoder_views(arena, prefix, self,
                                           arena.frozen[prefix + "pos_embed"].reshape(self.num_patches, -1), train)
        return v

    def forward_masks(self, x, masks):
        """All masks through one fused chain; returns a list with one [B, K_i, D] tensor per mask."""
        out, segs = self._run(x, masks)
        B = x.shape[0]
        return [out[s.row0:s.row0 + s.rows].view(B, s.S, self.embed_dim) for s in segs]

    def forward(self, x, masks=None):
        """x: fp32 clips [B,3,T,H,W] on the GPU; masks: None, an index tensor, or a list of [B,K] index tensors."""
        if masks is not None and not isinstance(masks, list):
            masks = [masks]
        self.interpolate_pos_encoding(x, self.pos_embed)
        outs = self.forward_masks(x, masks) if masks is not None else None
        if masks is None:
            out, segs = self._run(x, None)
            return out.view(x.shape[0], self.num_patches, self.embed_dim)
        return torch.cat(outs, dim=0)

    def _run(self, x, masks):
        hipmodule.require_gpu(x, "VisionTransformer.forward")
        x = x.contiguous().float()
        if masks is not None:
            masks = [m.contiguous() for m in masks]
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            return hipmodule.run_with_autograd(self, _enc_fwd, _enc_bwd, (x, masks))
        # inference: one C call per trunk (no saved activations, a private workspace) and, because nothing else shares the
        # GPU with this stream, the two-workgroups-per-CU GEMM (gemm4w.hip; see DESIGN.md section 6 for why training does not)
        ew = self._hip_views(train=False)
        # (ViT-L B=24 forward: 728.7 -> 761.1 clips/s; ViT-H, whose 1280 = 5 x 256 columns tile the big kernel exactly: 405.6 vs
        # 395.9, so the wide models keep the automatic selection -- tools/infer_bench.py)
        flags = INFER_GEMM_FLAGS if self.embed_dim <= 1024 else 0
        tag = f"infer{id(self)}:"
        if "_hip_ws_finalizer" not in self.__dict__:   # the workspace dies with the module
            import weakref
            from ...engine.chain import Workspace
            self.__dict__["_hip_ws_finalizer"] = weakref.finalize(self, Workspace.release, tag)
        out, segs, _ = encoder_forward(ew, x, masks, save=False, ws_tag=tag, gemm_flags=flags)
        return out, segs


INFER_GEMM_FLAGS = 0x100


def _enc_fwd(module, ew, args, diff):
    x, masks = args
    out, segs, saved = encoder_forward(ew, x, masks, save=True)
    return out, segs, (saved, segs)


def _enc_bwd(module, ew, ctx_saved, dout):
    saved, segs = ctx_saved
    encoder_backward(dout, saved, ew, segs, alpha=1.0)
    return None


def vit_tiny(patch_size=16, **kwargs):
    return VisionTransformer(patch_size=patch_size, embed_dim=192, depth=12, num_heads=3, mlp_ratio=4,
                             qkv_bias=True, norm_layer=partial(nn.LayerNorm, eps=1e-6), **kwargs)


def vit_small(patch_size=16, **kwargs):
    return VisionTransformer(patch_size=patch_size, embed_dim=384, depth=12, num_heads=6, mlp_ratio=4,
                             qkv_bias=True, norm_layer=partial(nn.LayerNorm, eps=1e-6), **kwargs)


def vit_base(patch_size=16, **kwargs):
    return VisionTransformer(patch_size=patch_size, embed_dim=768, depth=12, num_heads=12, mlp_ratio=4,
                             qkv_bias=True, norm_layer=partial(nn.LayerNorm, eps=1e-6), **kwargs)


def vit_large(patch_size=16, **kwargs):
    return VisionTransformer(patch_size=patch_size, embed_dim=1024, depth=24, num_heads=16, mlp_ratio=4,
                             qkv_bias=True, norm_layer=partial(nn.LayerNorm, eps=1e-6), **kwargs)


def vit_huge(patch_size=16, **kwargs):
    return VisionTransformer(patch_size=patch_size, embed_dim=1280, depth=32, num_heads=16, mlp_ratio=4,
                             qkv_bias=True, norm_layer=partial(nn.LayerNorm, eps=1e-6), **kwargs)


def vit_giant(patch_size=16, **kwargs):
    return VisionTransformer(patch_size=patch_size, embed_dim=1408, depth=40, num_heads=16, mlp_ratio=48 / 11,
                             qkv_bias=True, norm_layer=partial(nn.LayerNorm, eps=1e-6), **kwargs)


def vit_gigantic(patch_size=14, **kwargs):
    # the reference passes a misspelt `mpl_ratio` here (vision_transformer.py:293), so its effective mlp_ratio is
    # the default 4.0; kept for checkpoint compatibility
    return VisionTransformer(patch_size=patch_size, embed_dim=1664, depth=48, num_heads=16, mlp_ratio=4.0,
                             qkv_bias=True, norm_layer=partial(nn.LayerNorm, eps=1e-6), **kwargs)


VIT_EMBED_DIMS = {
    'vit_tiny': 192,
    'vit_small': 384,
    'vit_base': 768,
    'vit_large': 1024,
    'vit_huge': 1280,
    'vit_giant': 1408,
    'vit_gigantic': 1664,
}
