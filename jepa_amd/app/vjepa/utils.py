"""Model / optimizer construction and checkpoint loading with the reference's function names and arguments
(app/vjepa/utils.py:28-210), producing MI355X-native objects:

  init_video_model -> (MultiMaskWrapper(VisionTransformer), PredictorMultiMaskWrapper(VisionTransformerPredictor))
                      with the reference's parameter names and the same seed-for-seed initial weights
  init_opt         -> (optimizer, scaler, scheduler, wd_scheduler) where `optimizer` is the fused
                      AdamW+EMA+bf16-recast Trainer (param_groups / state_dict / load_state_dict compatible)
"""
import logging
import sys

import torch

from ...engine.step import Trainer
from ...src.models import predictor as vit_pred
from ...src.models import vision_transformer as video_vit
from ...src.models.utils.multimask import MultiMaskWrapper, PredictorMultiMaskWrapper
from ...src.utils.schedulers import CosineWDSchedule, WarmupCosineSchedule
from ...src.utils.tensors import trunc_normal_

logging.basicConfig(stream=sys.stdout, level=logging.INFO)
logger = logging.getLogger()


def load_checkpoint(r_path, encoder, predictor, target_encoder, opt, scaler):
    """Restore the five state dicts written by train.save_checkpoint; on any failure log and restart from
    epoch 0, like the reference (utils.py:28-83)."""
    epoch = 0
    try:
        checkpoint = torch.load(r_path, map_location=torch.device('cpu'))
        epoch = checkpoint['epoch']
        for key, module in (('encoder', encoder), ('predictor', predictor), ('target_encoder', target_encoder)):
            if module is None:
                continue
            sd = {k[len('module.'):] if k.startswith('module.') else k: v for k, v in checkpoint[key].items()}
            msg = module.load_state_dict(sd)
            logger.info(f'loaded pretrained {key} from epoch {epoch} with msg: {msg}')
        opt.load_state_dict(checkpoint['opt'])
        if hasattr(opt, 'sync_shadows'):
            opt.sync_shadows()
        logger.info(f'loaded optimizers from epoch {epoch}; read-path: {r_path}')
        del checkpoint
    except Exception as e:
        logger.info(f'Encountered exception when loading checkpoint {e}')
        epoch = 0
    return encoder, predictor, target_encoder, opt, scaler, epoch


def init_video_model(device, patch_size=16, num_frames=16, tubelet_size=2, model_name='vit_base', crop_size=224,
                     pred_depth=6, pred_embed_dim=384, uniform_power=False, use_mask_tokens=False,
                     num_mask_tokens=2, zero_init_mask_tokens=True, use_sdpa=False):
    encoder = video_vit.__dict__[model_name](img_size=crop_size, patch_size=patch_size, num_frames=num_frames,
                                             tubelet_size=tubelet_size, uniform_power=uniform_power,
                                             use_sdpa=use_sdpa)
    encoder = MultiMaskWrapper(encoder)
    predictor = vit_pred.__dict__['vit_predictor'](
        img_size=crop_size, use_mask_tokens=use_mask_tokens, patch_size=patch_size, num_frames=num_frames,
        tubelet_size=tubelet_size, embed_dim=encoder.backbone.embed_dim, predictor_embed_dim=pred_embed_dim,
        depth=pred_depth, num_heads=encoder.backbone.num_heads, uniform_power=uniform_power,
        num_mask_tokens=num_mask_tokens, zero_init_mask_tokens=zero_init_mask_tokens, use_sdpa=use_sdpa)
    predictor = PredictorMultiMaskWrapper(predictor)

    # The reference re-initialises every Linear / LayerNorm after construction (utils.py:127-140), which also
    # undoes the depth-wise rescale of proj/fc2; reproduced so that equal seeds give equal weights.
    for model in (encoder, predictor):
        for m in model.modules():
            if isinstance(m, torch.nn.Linear):
                trunc_normal_(m.weight, std=0.02)
                if m.bias is not None:
                    torch.nn.init.constant_(m.bias, 0)
            elif isinstance(m, torch.nn.LayerNorm):
                torch.nn.init.constant_(m.bias, 0)
                torch.nn.init.constant_(m.weight, 1.0)
    encoder.to(device)
    predictor.to(device)

    def count_parameters(model):
        return sum(p.numel() for p in model.parameters() if p.requires_grad)

    logger.info(f'Encoder number of parameters: {count_parameters(encoder)}')
    logger.info(f'Predictor number of parameters: {count_parameters(predictor)}')
    return encoder, predictor


class _NoScaler:
    """Stand-in for torch.cuda.amp.GradScaler: bf16 shares fp32's exponent range, so loss scaling is the
    identity; state_dict() keeps checkpoints loadable by the reference."""

    def state_dict(self):
        return {"scale": 1.0, "growth_factor": 2.0, "backoff_factor": 0.5, "growth_interval": 2000,
                "_growth_tracker": 0}

    def load_state_dict(self, sd):
        return None


def init_opt(encoder, predictor, iterations_per_epoch, start_lr, ref_lr, warmup, num_epochs, wd=1e-6,
             final_wd=1e-6, final_lr=0.0, mixed_precision=False, ipe_scale=1.25, betas=(0.9, 0.999), eps=1e-8,
             zero_init_bias_wd=True, target_encoder=None, loss_exp=1.0, reg_coeff=0.0, clip_grad=None,
             world_size=1, device=None, micro_batch=None, overlap_update=False):
    """Same schedules and parameter grouping as the reference; the returned optimizer is the fused Trainer.
    Extra keyword arguments (target_encoder, loss_exp, reg_coeff, clip_grad, world_size) configure the step; gradient
    non-finite checks (GradScaler's skip semantics) are always on and device-side."""
    if target_encoder is None:
        # reference-compatible call (app/vjepa/utils.py:156-170 has no such argument): the EMA update is fused into the
        # optimizer kernel, so the optimizer owns the target.  Built exactly as train.py:276-277 builds it (a deep copy of
        # the freshly initialised encoder, frozen) and exposed as `optimizer.target_encoder`; a caller that keeps its own
        # copy.deepcopy(encoder) must use this one instead -- the fused update writes only here.
        import copy
        target_encoder = copy.deepcopy(encoder)
        for p in target_encoder.parameters():
            p.requires_grad = False
        import warnings
        warnings.warn('jepa_amd init_opt: no target_encoder= given -- the EMA target is optimizer.target_encoder (a deep copy of the '
                      'encoder, updated by the fused optimizer kernel); a target copy kept and EMA-updated by the caller would be a '
                      'SECOND target that this optimizer never writes', stacklevel=2)
    optimizer = Trainer(encoder, predictor, target_encoder, loss_exp=loss_exp, reg_coeff=reg_coeff, betas=betas,
                        eps=eps, clip_grad=clip_grad, world_size=world_size, device=device, micro_batch=micro_batch,
                        overlap_update=overlap_update)
    scheduler = WarmupCosineSchedule(optimizer, warmup_steps=int(warmup * iterations_per_epoch), start_lr=start_lr,
                                     ref_lr=ref_lr, final_lr=final_lr,
                                     T_max=int(ipe_scale * num_epochs * iterations_per_epoch))
    wd_scheduler = CosineWDSchedule(optimizer, ref_wd=wd, final_wd=final_wd,
                                    T_max=int(ipe_scale * num_epochs * iterations_per_epoch))
    scaler = _NoScaler() if mixed_precision else None
    return optimizer, scaler, scheduler, wd_scheduler
