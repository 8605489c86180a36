"""V-JEPA predictor with the reference's constructor, attributes and state-dict names
(src/models/predictor.py:23-246), computing on MI355X through the jepa_amd HIP kernels.

    pred = vit_predictor(img_size=224, patch_size=16, num_frames=16, tubelet_size=2, embed_dim=1024,
                         predictor_embed_dim=384, depth=12, num_heads=16, uniform_power=True,
                         use_mask_tokens=True, num_mask_tokens=2, zero_init_mask_tokens=True)
    z = pred(ctxt, tgt, masks_ctxt, masks_tgt, mask_index=i)   # [B, K_pred, embed_dim]
"""
import math
from functools import partial

import torch
import torch.nn as nn

from ...engine import hipmodule
from ...engine.layers import Seg, predictor_backward, predictor_forward
from ..utils.tensors import trunc_normal_
from .utils.modules import Block
from .utils.pos_embs import get_3d_sincos_pos_embed


class VisionTransformerPredictor(nn.Module, hipmodule.HipModule):
    """ Vision Transformer predictor (video) """

    def __init__(self, img_size=224, patch_size=16, num_frames=1, tubelet_size=2, embed_dim=768,
                 predictor_embed_dim=384, depth=6, num_heads=12, mlp_ratio=4.0, qkv_bias=True, qk_scale=None,
                 drop_rate=0.0, attn_drop_rate=0.0, norm_layer=nn.LayerNorm, init_std=0.02, uniform_power=False,
                 use_mask_tokens=False, num_mask_tokens=2, zero_init_mask_tokens=True, **kwargs):
        super().__init__()
        if not use_mask_tokens:
            raise NotImplementedError("the diffusion-noise branch (use_mask_tokens=False, predictor.py:154-172,203-205) "
                                      "is unused by every V-JEPA pretraining config")
        if num_frames <= 1:
            raise NotImplementedError("image predictors are outside the video pretraining step")
        self.embed_dim = embed_dim
        self.num_heads = num_heads
        self.predictor_embed = nn.Linear(embed_dim, predictor_embed_dim, bias=True)
        self.num_mask_tokens = num_mask_tokens
        self.mask_tokens = nn.ParameterList([nn.Parameter(torch.zeros(1, 1, predictor_embed_dim))
                                             for _ in range(num_mask_tokens)])
        self.input_size = img_size
        self.patch_size = patch_size
        self.num_frames = num_frames
        self.tubelet_size = tubelet_size
        self.is_video = True
        grid_size = img_size // patch_size
        grid_depth = num_frames // tubelet_size
        self.num_patches = grid_depth * grid_size * grid_size
        self.uniform_power = uniform_power
        self.predictor_pos_embed = nn.Parameter(torch.zeros(1, self.num_patches, predictor_embed_dim),
                                                requires_grad=False)
        self.predictor_blocks = nn.ModuleList([
            Block(dim=predictor_embed_dim, num_heads=num_heads, mlp_ratio=mlp_ratio, qkv_bias=qkv_bias,
                  qk_scale=qk_scale, drop=drop_rate, act_layer=nn.GELU, attn_drop=attn_drop_rate,
                  grid_size=grid_size, grid_depth=grid_depth, norm_layer=norm_layer) for _ in range(depth)])
        self.predictor_norm = norm_layer(predictor_embed_dim)
        self.predictor_proj = nn.Linear(predictor_embed_dim, embed_dim, bias=True)
        sincos = get_3d_sincos_pos_embed(predictor_embed_dim, grid_size, grid_depth, cls_token=False,
                                         uniform_power=uniform_power)
        self.predictor_pos_embed.data.copy_(torch.from_numpy(sincos).float().unsqueeze(0))
        self.init_std = init_std
        if not zero_init_mask_tokens:
            for mt in self.mask_tokens:
                trunc_normal_(mt, std=init_std)
        self.apply(self._init_weights)
        self._rescale_blocks()

    def _init_weights(self, m):
        if isinstance(m, nn.Linear):
            trunc_normal_(m.weight, std=self.init_std)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, nn.LayerNorm):
            nn.init.constant_(m.bias, 0)
            nn.init.constant_(m.weight, 1.0)

    def _rescale_blocks(self):
        for layer_id, layer in enumerate(self.predictor_blocks):
            layer.attn.proj.weight.data.div_(math.sqrt(2.0 * (layer_id + 1)))
            layer.mlp.fc2.weight.data.div_(math.sqrt(2.0 * (layer_id + 1)))

    # ---- compute ------------------------------------------------------------------------------------------
    def _hip_views(self, train):
        from ...engine.weights import predictor_views
        arena, prefix = self._hip_arena(train)
        pos = arena.frozen[prefix + "predictor_pos_embed"].reshape(self.num_patches, -1)
        return predictor_views(arena, prefix, self, pos, train)

    def forward_masks(self, ctxt, tgt, masks_ctxt, masks_tgt, first_mask_index=0):
        """One fused chain over all (context, mask) pairs; returns a list of [B, K_pred_i, embed_dim]."""
        assert len(ctxt) == len(masks_ctxt) == len(masks_tgt), 'need one context tensor per mask pair'
        B = ctxt[0].shape[0]
        D = self.embed_dim
        for c in ctxt:
            hipmodule.require_gpu(c, "VisionTransformerPredictor.forward")
        z = torch.cat([c.reshape(-1, D) for c in ctxt], dim=0) if len(ctxt) > 1 else ctxt[0].reshape(-1, D)
        enc_segs, r = [], 0
        for c in ctxt:
            enc_segs.append(Seg(r, B, c.shape[1]))
            r += B * c.shape[1]
        masks_ctxt = [m.contiguous() for m in masks_ctxt]
        masks_tgt = [m.contiguous() for m in masks_tgt]
        args = (enc_segs, masks_ctxt, masks_tgt, first_mask_index)
        if torch.is_grad_enabled() and (z.requires_grad or any(p.requires_grad for p in self.parameters())):
            zhat, tsegs = hipmodule.run_with_autograd(self, _pred_fwd, _pred_bwd, args, diff_inputs=(z,))
        else:
            pw = self._hip_views(train=False)
            zhat, tsegs, _ = _pred_call(pw, z, args, save=False)
        return [zhat[s.row0:s.row0 + s.rows].view(B, s.S, D) for s in tsegs]

    def forward(self, ctxt, tgt, masks_ctxt, masks_tgt, mask_index=1):
        """ctxt: [B*len(masks_ctxt), K_ctx, D] context tokens; masks_*: index tensors (or lists of them)."""
        assert (masks_ctxt is not None) and (masks_tgt is not None), 'Cannot run predictor without mask indices'
        if not isinstance(masks_ctxt, list):
            masks_ctxt = [masks_ctxt]
        if not isinstance(masks_tgt, list):
            masks_tgt = [masks_tgt]
        if len(masks_ctxt) != 1 or len(masks_tgt) != 1:
            raise NotImplementedError("forward() takes one (context, target) mask pair per call, as issued by "
                                      "PredictorMultiMaskWrapper; use forward_masks for several pairs")
        return self.forward_masks([ctxt], [tgt], masks_ctxt, masks_tgt, first_mask_index=mask_index)[0]


def _rot(pw, first):
    """View of the weights whose mask-token list starts at `first` (mask_index % num_mask_tokens, predictor.py:206)."""
    n = len(pw.mask_tokens)
    if first % n == 0:
        return pw
    return type(pw)(**{**pw.__dict__, "mask_tokens": [pw.mask_tokens[(first + i) % n] for i in range(n)],
                       "g_mask_tokens": [pw.g_mask_tokens[(first + i) % n] for i in range(n)]})


def _pred_call(pw, z, args, save):
    enc_segs, masks_ctxt, masks_tgt, first = args
    return predictor_forward(_rot(pw, first), z.to(torch.bfloat16).contiguous(), enc_segs, masks_ctxt, masks_tgt, save)


def _pred_fwd(module, pw, args, diff):
    zhat, tsegs, saved = _pred_call(pw, diff[0], args, save=True)
    return zhat, tsegs, (saved, args)


def _pred_bwd(module, pw, ctx_saved, dout):
    saved, args = ctx_saved
    dz = predictor_backward(dout.to(torch.bfloat16).contiguous(), saved, _rot(pw, args[3]), args[0], alpha=1.0)
    return [dz]


def vit_predictor(**kwargs):
    return VisionTransformerPredictor(mlp_ratio=4, qkv_bias=True, norm_layer=partial(nn.LayerNorm, eps=1e-6),
                                      **kwargs)
