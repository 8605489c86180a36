"""Builders shared by the GPU step tests: the same seeded model as (a) a jepa_amd Trainer on the GPU and (b) the flat
weight dict the CPU oracle steps."""
import copy
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

TINY = dict(model_name="vit_tiny", crop=64, frames=8, patch=16, tubelet=2, pred_depth=2, pred_dim=96, embed_dim=192,
            depth=12, heads=3, num_patches=64)
VITL = dict(model_name="vit_large", crop=224, frames=16, patch=16, tubelet=2, pred_depth=12, pred_dim=384,
            embed_dim=1024, depth=24, heads=16, num_patches=1568)
VITH = dict(model_name="vit_huge", crop=224, frames=16, patch=16, tubelet=2, pred_depth=12, pred_dim=384,
            embed_dim=1280, depth=32, heads=16, num_patches=1568)
VITL_MASKS = [dict(aspect_ratio=(0.75, 1.5), num_blocks=8, spatial_scale=(0.15, 0.15), temporal_scale=(1.0, 1.0)),
              dict(aspect_ratio=(0.75, 1.5), num_blocks=2, spatial_scale=(0.7, 0.7), temporal_scale=(1.0, 1.0))]
TINY_MASKS = [dict(aspect_ratio=(0.75, 1.5), num_blocks=4, spatial_scale=(0.15, 0.15), temporal_scale=(1.0, 1.0)),
              dict(aspect_ratio=(0.75, 1.5), num_blocks=1, spatial_scale=(0.6, 0.6), temporal_scale=(0.5, 1.0))]


def oracle_cfg(m, n_masks):
    return dict(embed_dim=m["embed_dim"], depth=m["depth"], heads=m["heads"], pred_dim=m["pred_dim"],
                pred_depth=m["pred_depth"], num_mask_tokens=n_masks, patch=m["patch"], tubelet=m["tubelet"],
                num_patches=m["num_patches"])


def oracle_state(enc, pred):
    ow_enc = {k[len("backbone."):]: v.detach().clone().cpu() for k, v in enc.state_dict().items()}
    ow_pred = {k[len("backbone."):]: v.detach().clone().cpu() for k, v in pred.state_dict().items()}
    return dict(enc=ow_enc, pred=ow_pred, tgt={k: v.clone() for k, v in ow_enc.items()}, opt={})


def build_models(m, n_masks, seed=0, perturb_small=False):
    """(encoder, predictor) on CPU, seeded like bench.py / the reference's init_video_model call."""
    from jepa_amd.app.vjepa.utils import init_video_model
    torch.manual_seed(seed)
    enc, pred = init_video_model(device="cpu", patch_size=m["patch"], num_frames=m["frames"], tubelet_size=m["tubelet"],
                                 model_name=m["model_name"], crop_size=m["crop"], pred_depth=m["pred_depth"],
                                 pred_embed_dim=m["pred_dim"], uniform_power=True, use_mask_tokens=True,
                                 num_mask_tokens=n_masks, zero_init_mask_tokens=True)
    if perturb_small:   # non-trivial biases / LayerNorm affine / mask tokens (they are 0 / 1 at init)
        g = torch.Generator().manual_seed(5)
        with torch.no_grad():
            for mod in (enc, pred):
                for n, p in mod.named_parameters():
                    if p.requires_grad and (p.dim() == 1 or "mask_tokens" in n):
                        p.add_(0.02 * torch.randn(p.shape, generator=g))
    return enc, pred


def build_trainer(m, n_masks, device="cuda", seed=0, perturb_small=False, **trainer_kw):
    """-> (trainer, oracle_state, encoder, predictor, target_encoder); the oracle state is cloned BEFORE the modules'
    storage moves into the trainer's arenas."""
    from jepa_amd.engine.step import Trainer
    enc, pred = build_models(m, n_masks, seed, perturb_small)
    state = oracle_state(enc, pred)
    tgt = copy.deepcopy(enc)
    for p in tgt.parameters():
        p.requires_grad = False
    enc.to(device), pred.to(device), tgt.to(device)
    tr = Trainer(enc, pred, tgt, device=device, **trainer_kw)
    return tr, state, enc, pred, tgt


def draw_batch(gens, B, m, clip_seed, mask_seed):
    clips = torch.randn(B, 3, m["frames"], m["crop"], m["crop"], generator=torch.Generator().manual_seed(clip_seed))
    torch.manual_seed(mask_seed)
    me, mp = zip(*[g(B) for g in gens])
    return clips, list(me), list(mp)


def to_dev(clips, me, mp, device="cuda"):
    return clips.to(device), [x.to(device) for x in me], [x.to(device) for x in mp]
