"""V-JEPA pretraining loop with the reference's entry point and YAML schema (app/vjepa/train.py:66-586):

    from jepa_amd.app.vjepa.train import main
    main(args_dict_from_yaml, resume_preempt=False)

Same config keys, schedules, momentum generator, CSV columns, checkpoint dictionary and state-dict key names
(`module.backbone.*`, as written by the reference's DDP-wrapped modules) -- but `train_step` is the fused MI355X
step of jepa_amd.engine.step.Trainer (hand-written gfx950 kernels, no autograd, RCCL gradient all-reduce
overlapped with backward) instead of autocast + DDP + torch.optim.AdamW + a per-tensor EMA loop.
"""
import copy
import os
import time

import numpy as np
import torch
import torch.multiprocessing as mp

from ...engine import dp
from ...engine.input import DevicePrefetcher
from ...src.datasets.data_manager import init_data
from ...src.masks.multiblock3d import MaskCollator as MB3DMaskCollator
from ...src.masks.random_tube import MaskCollator as TubeMaskCollator
from ...src.utils.distributed import AllReduce, init_distributed
from ...src.utils.logging import AverageMeter, CSVLogger, adamw_logger, get_logger, gpu_timer, grad_logger
from .utils import init_opt, init_video_model, load_checkpoint

log_timings = True
log_freq = 10
checkpoint_freq = 1

_GLOBAL_SEED = 0
np.random.seed(_GLOBAL_SEED)
torch.manual_seed(_GLOBAL_SEED)

logger = get_logger(__name__)


_REQUIRED = object()
# (attribute, YAML section, key, default, cast) -- the reference's config surface (app/vjepa/train.py:71-157) plus one
# extension: optimization.micro_batch = clips per micro-batch inside one step (the 3072-clip ViT-H recipe puts 384 clips
# on a GPU; their saved activations do not fit, so the step walks them with gradient accumulation -- same sums, the
# collator still draws masks for the whole batch).
_SCHEMA = (
    ('load_model', 'meta', 'load_checkpoint', None, None), ('r_file', 'meta', 'read_checkpoint', None, None),
    ('seed', 'meta', 'seed', _GLOBAL_SEED, None), ('save_every_freq', 'meta', 'save_every_freq', -1, None),
    ('skip_batches', 'meta', 'skip_batches', -1, None), ('use_sdpa', 'meta', 'use_sdpa', False, None),
    ('which_dtype', 'meta', 'dtype', _REQUIRED, None),
    ('model_name', 'model', 'model_name', _REQUIRED, None), ('pred_depth', 'model', 'pred_depth', _REQUIRED, None),
    ('pred_embed_dim', 'model', 'pred_embed_dim', _REQUIRED, None), ('uniform_power', 'model', 'uniform_power', True, None),
    ('use_mask_tokens', 'model', 'use_mask_tokens', True, None),
    ('zero_init_mask_tokens', 'model', 'zero_init_mask_tokens', True, None),
    ('dataset_type', 'data', 'dataset_type', 'videodataset', None), ('mask_type', 'data', 'mask_type', 'multiblock3d', None),
    ('dataset_paths', 'data', 'datasets', [], None), ('datasets_weights', 'data', 'datasets_weights', None, None),
    ('batch_size', 'data', 'batch_size', _REQUIRED, None), ('num_clips', 'data', 'num_clips', _REQUIRED, None),
    ('num_frames', 'data', 'num_frames', _REQUIRED, None), ('tubelet_size', 'data', 'tubelet_size', _REQUIRED, None),
    ('sampling_rate', 'data', 'sampling_rate', None, None), ('duration', 'data', 'clip_duration', None, None),
    ('crop_size', 'data', 'crop_size', 224, None), ('patch_size', 'data', 'patch_size', _REQUIRED, None),
    ('pin_mem', 'data', 'pin_mem', False, None), ('num_workers', 'data', 'num_workers', 1, None),
    ('filter_short_videos', 'data', 'filter_short_videos', False, None),
    ('decode_one_clip', 'data', 'decode_one_clip', True, None),
    ('loss_exp', 'loss', 'loss_exp', _REQUIRED, None), ('reg_coeff', 'loss', 'reg_coeff', _REQUIRED, None),
    ('ipe', 'optimization', 'ipe', None, None), ('ipe_scale', 'optimization', 'ipe_scale', 1.0, None),
    ('clip_grad', 'optimization', 'clip_grad', None, None), ('wd', 'optimization', 'weight_decay', _REQUIRED, float),
    ('final_wd', 'optimization', 'final_weight_decay', _REQUIRED, float), ('num_epochs', 'optimization', 'epochs', _REQUIRED, None),
    ('warmup', 'optimization', 'warmup', _REQUIRED, None), ('start_lr', 'optimization', 'start_lr', _REQUIRED, None),
    ('lr', 'optimization', 'lr', _REQUIRED, None), ('final_lr', 'optimization', 'final_lr', _REQUIRED, None),
    ('ema', 'optimization', 'ema', _REQUIRED, None), ('betas', 'optimization', 'betas', (0.9, 0.999), None),
    ('eps', 'optimization', 'eps', 1.e-8, None), ('micro_batch', 'optimization', 'micro_batch', None, None),
    ('overlap_update', 'optimization', 'overlap_update', True, None),
    ('folder', 'logging', 'folder', _REQUIRED, None), ('tag', 'logging', 'write_tag', _REQUIRED, None),
)


def _parse_config(args):
    """YAML dict -> namespace.  Like the reference's chain of .get() calls, a key without a default that is missing reads
    as None (and fails where it is first used); `cast` mirrors the reference's float() on the weight-decay keys."""
    import types
    cfg = types.SimpleNamespace()
    for attr, section, key, default, cast in _SCHEMA:
        sec = args.get(section) or {}
        val = sec.get(key, None if default is _REQUIRED else default)
        setattr(cfg, attr, cast(val) if (cast is not None and val is not None) else val)
    return cfg


def _with_module_prefix(sd):
    return {"module." + k: v for k, v in sd.items()}


def main(args, resume_preempt=False):
    # config: the reference's YAML schema (app/vjepa/train.py:71-157) as a table -- see _SCHEMA below
    cfg = _parse_config(args)
    cfg.load_model = cfg.load_model or resume_preempt
    logger.info(f'which_dtype={cfg.which_dtype!r}')
    if str(cfg.which_dtype).lower() != 'bfloat16':
        raise NotImplementedError("meta.dtype must be bfloat16: the MI355X path computes in bf16 MFMA with fp32 "
                                  "accumulation and fp32 master weights (every shipped pretrain config uses bfloat16)")
    if cfg.datasets_weights is not None and len(cfg.datasets_weights) != len(cfg.dataset_paths):
        raise AssertionError('Must have one sampling weight specified for each dataset')
    cfgs_mask = args.get('mask')
    mixed_precision = True
    seed, ipe, num_epochs, warmup, clip_grad = cfg.seed, cfg.ipe, cfg.num_epochs, cfg.warmup, cfg.clip_grad
    batch_size, folder, tag = cfg.batch_size, cfg.folder, cfg.tag

    np.random.seed(seed)
    torch.manual_seed(seed)
    try:
        mp.set_start_method('spawn')
    except Exception:
        pass

    world_size, rank = init_distributed()
    logger.info(f'Initialized (rank/world-size) {rank}/{world_size}')
    if not torch.cuda.is_available():
        raise RuntimeError("jepa_amd.app.vjepa.train needs an MI355X: the step runs only in libvjepa_hip.so")
    device = torch.device('cuda', torch.cuda.current_device())

    os.makedirs(folder, exist_ok=True)
    log_file = os.path.join(folder, f'{tag}_r{rank}.csv')
    latest_path = os.path.join(folder, f'{tag}-latest.pth.tar')
    load_path = None
    if cfg.load_model:
        load_path = os.path.join(folder, cfg.r_file) if cfg.r_file is not None else latest_path
        if not os.path.exists(load_path):
            load_path, cfg.load_model = None, False

    csv_logger = CSVLogger(log_file, ('%d', 'epoch'), ('%d', 'itr'), ('%.5f', 'loss'), ('%.5f', 'loss-jepa'),
                           ('%.5f', 'reg-loss'), ('%.5f', 'enc-grad-norm'), ('%.5f', 'pred-grad-norm'),
                           ('%d', 'gpu-time(ms)'), ('%d', 'wall-time(ms)'))

    encoder, predictor = init_video_model(
        uniform_power=cfg.uniform_power, use_mask_tokens=cfg.use_mask_tokens, num_mask_tokens=len(cfgs_mask),
        zero_init_mask_tokens=cfg.zero_init_mask_tokens, device='cpu', patch_size=cfg.patch_size, num_frames=cfg.num_frames,
        tubelet_size=cfg.tubelet_size, model_name=cfg.model_name, crop_size=cfg.crop_size, pred_depth=cfg.pred_depth,
        pred_embed_dim=cfg.pred_embed_dim, use_sdpa=cfg.use_sdpa)
    target_encoder = copy.deepcopy(encoder)
    for p in target_encoder.parameters():
        p.requires_grad = False

    collator_cls = MB3DMaskCollator if cfg.mask_type == 'multiblock3d' else TubeMaskCollator
    mask_collator = collator_cls(crop_size=cfg.crop_size, num_frames=cfg.num_frames, patch_size=cfg.patch_size,
                                 tubelet_size=cfg.tubelet_size, cfgs_mask=cfgs_mask)

    (unsupervised_loader, unsupervised_sampler) = init_data(
        data=cfg.dataset_type, root_path=cfg.dataset_paths, batch_size=batch_size, training=True, clip_len=cfg.num_frames,
        frame_sample_rate=cfg.sampling_rate, filter_short_videos=cfg.filter_short_videos, decode_one_clip=cfg.decode_one_clip,
        duration=cfg.duration, num_clips=cfg.num_clips, transform=None, datasets_weights=cfg.datasets_weights,
        collator=mask_collator, num_workers=cfg.num_workers, world_size=world_size, pin_mem=cfg.pin_mem, rank=rank,
        log_dir=None, crop_size=cfg.crop_size)
    try:
        _dlen = len(unsupervised_loader)
    except Exception:
        _dlen = unsupervised_loader.num_batches
    if ipe is None:
        ipe = _dlen
    logger.info(f'iterations per epoch/dataest length: {ipe}/{_dlen}')

    optimizer, scaler, scheduler, wd_scheduler = init_opt(
        encoder=encoder, predictor=predictor, target_encoder=target_encoder, wd=cfg.wd, final_wd=cfg.final_wd,
        start_lr=cfg.start_lr, ref_lr=cfg.lr, final_lr=cfg.final_lr, iterations_per_epoch=ipe, warmup=warmup,
        num_epochs=num_epochs, ipe_scale=cfg.ipe_scale, mixed_precision=mixed_precision, betas=cfg.betas, eps=cfg.eps,
        loss_exp=cfg.loss_exp, reg_coeff=cfg.reg_coeff, clip_grad=clip_grad, world_size=world_size, device=device,
        micro_batch=cfg.micro_batch, overlap_update=bool(cfg.overlap_update))
    trainer = optimizer
    dp.broadcast_parameters(trainer.arena, trainer.tarena)   # DDP's one-time parameter sync (train.py:295-297)
    if world_size > 1:
        trainer.sync_shadows()

    momentum_scheduler = (cfg.ema[0] + i * (cfg.ema[1] - cfg.ema[0]) / (ipe * num_epochs * cfg.ipe_scale)
                          for i in range(int(ipe * num_epochs * cfg.ipe_scale) + 1))

    start_epoch = 0
    if cfg.load_model or os.path.exists(latest_path):
        # like the reference (train.py:307-320): with meta.load_checkpoint unset, load_path is None, the load fails, is
        # logged and training starts at epoch 0 -- an existing `-latest` file in a reused folder is NOT auto-resumed
        encoder, predictor, target_encoder, optimizer, scaler, start_epoch = load_checkpoint(
            r_path=load_path, encoder=encoder, predictor=predictor, target_encoder=target_encoder, opt=optimizer,
            scaler=scaler)
        for _ in range(start_epoch * ipe):   # replay schedules and the mask counter (train.py:322-326)
            scheduler.step()
            wd_scheduler.step()
            next(momentum_scheduler)
            mask_collator.step()

    def save_checkpoint(epoch, path):
        if rank != 0:
            return
        trainer.sync_update()   # optimization.overlap_update: the last step's fused update may still be in flight on its own stream
        save_dict = {
            'encoder': _with_module_prefix(encoder.state_dict()),
            'predictor': _with_module_prefix(predictor.state_dict()),
            'opt': optimizer.state_dict(),
            'scaler': None if scaler is None else scaler.state_dict(),
            'target_encoder': _with_module_prefix(target_encoder.state_dict()),
            'epoch': epoch, 'loss': loss_meter.avg, 'batch_size': batch_size, 'world_size': world_size, 'lr': cfg.lr,
        }
        try:
            torch.save(save_dict, path)
        except Exception as e:
            logger.info(f'Encountered exception when saving checkpoint: {e}')

    logger.info('Initializing loader...')
    loader = iter(unsupervised_loader)
    if cfg.skip_batches > 0:
        unsupervised_sampler.set_epoch(start_epoch)
        for itr in range(cfg.skip_batches):
            try:
                next(loader)
            except Exception:
                loader = iter(unsupervised_loader)
                next(loader)

    def fetch_host_batch():
        """next(loader) with the reference's refresh-on-exhaustion (train.py:372-381); runs one step ahead."""
        nonlocal loader
        try:
            udata, masks_enc, masks_pred = next(loader)
        except Exception:
            logger.info('Exhausted data loaders. Refreshing...')
            loader = iter(unsupervised_loader)
            udata, masks_enc, masks_pred = next(loader)
        assert len(masks_enc) == len(masks_pred), 'Currently require num encoder masks = num predictor masks'
        return udata[0], masks_enc, masks_pred

    # load_clips (train.py:391-408) one batch ahead: pinned staging + copy stream, see engine/input.py
    prefetcher = DevicePrefetcher(fetch_host_batch, device, batch_size=batch_size, num_clips=cfg.num_clips)

    for epoch in range(start_epoch, num_epochs):
        logger.info('Epoch %d' % (epoch + 1))
        unsupervised_sampler.set_epoch(epoch)
        loss_meter, input_var_meter, input_var_min_meter = AverageMeter(), AverageMeter(), AverageMeter()
        jepa_loss_meter, reg_loss_meter = AverageMeter(), AverageMeter()
        mask_meters = [AverageMeter() for _ in range(len(cfgs_mask))]
        gpu_time_meter, wall_time_meter = AverageMeter(), AverageMeter()

        for itr in range(ipe):
            itr_start_time = time.time()
            clips, masks_enc, masks_pred = prefetcher.next(lookahead=(itr != ipe - 1))   # never across an epoch boundary
            for _i, m in enumerate(mask_meters):
                m.update(masks_enc[_i][0].size(-1))

            def train_step():
                _new_lr = scheduler.step()
                _new_wd = wd_scheduler.step()
                m = next(momentum_scheduler)
                out = trainer.train_step(clips, masks_enc, masks_pred, lr=_new_lr, wd=_new_wd, ema=m,
                                         clip_now=(epoch > warmup) and (clip_grad is not None))
                # per-tensor gradient / moment statistics (train.py:476-481) only when their log line is due: one
                # launch over the arenas instead of ~1000 float() syncs per step
                stats = None
                if itr % log_freq == 0:
                    stats = (grad_logger(trainer, 'enc'), grad_logger(trainer, 'pred'), adamw_logger(trainer))
                return (out.loss, out.loss_jepa, out.loss_reg, _new_lr, _new_wd, out.grad_norms, stats)

            (loss, loss_jepa, loss_reg, _new_lr, _new_wd, grad_norms, stats), gpu_etime_ms = gpu_timer(train_step)
            iter_elapsed_time_ms = (time.time() - itr_start_time) * 1000.
            loss_meter.update(loss)
            if itr % log_freq == 0:   # input statistics only when they are printed (one fused reduction each)
                flat = clips.view(clips.shape[0], -1)
                input_var = float(AllReduce.apply(flat.var(dim=1).mean(dim=0)))
                input_var_min = float(AllReduce.apply(torch.min(flat.var(dim=1))))
                input_var_meter.update(input_var)
                input_var_min_meter.update(input_var_min)
            jepa_loss_meter.update(loss_jepa)
            reg_loss_meter.update(loss_reg)
            gpu_time_meter.update(gpu_etime_ms)
            wall_time_meter.update(iter_elapsed_time_ms)

            csv_logger.log(epoch + 1, itr, loss, loss_jepa, loss_reg, grad_norms[0], grad_norms[1], gpu_etime_ms,
                           iter_elapsed_time_ms)
            if (itr % log_freq == 0) or np.isnan(loss) or np.isinf(loss):
                logger.info('[%d, %5d] loss: %.3f | p%.3f r%.3f | input_var: %.3f %.3f | masks: %s '
                            '[wd: %.2e] [lr: %.2e] [mem: %.2e] [gpu: %.1f ms][wall: %.1f ms]'
                            % (epoch + 1, itr, loss_meter.avg, jepa_loss_meter.avg, reg_loss_meter.avg,
                               input_var_meter.avg, input_var_min_meter.avg,
                               '[' + ', '.join(['%.1f' % m.avg for m in mask_meters]) + ']', _new_wd, _new_lr,
                               torch.cuda.max_memory_allocated() / 1024.0 ** 2, gpu_time_meter.avg,
                               wall_time_meter.avg))
                if stats is not None:
                    grad_stats, grad_stats_pred, optim_stats = stats
                    logger.info('[%d, %5d] first moment: %.2e [%.2e %.2e] second moment: %.2e [%.2e %.2e]'
                                % (epoch + 1, itr, optim_stats.get('exp_avg').avg, optim_stats.get('exp_avg').min,
                                   optim_stats.get('exp_avg').max, optim_stats.get('exp_avg_sq').avg,
                                   optim_stats.get('exp_avg_sq').min, optim_stats.get('exp_avg_sq').max))
                    logger.info('[%d, %5d] enc_grad_stats: f/l[%.2e %.2e] mn/mx(%.2e, %.2e) %.2e'
                                % (epoch + 1, itr, grad_stats.first_layer, grad_stats.last_layer, grad_stats.min,
                                   grad_stats.max, grad_norms[0]))
                    logger.info('[%d, %5d] pred_grad_stats: f/l[%.2e %.2e] mn/mx(%.2e, %.2e) %.2e'
                                % (epoch + 1, itr, grad_stats_pred.first_layer, grad_stats_pred.last_layer,
                                   grad_stats_pred.min, grad_stats_pred.max, grad_norms[1]))
            assert not np.isnan(loss), 'loss is nan'

        logger.info('avg. loss %.3f' % loss_meter.avg)
        if epoch % checkpoint_freq == 0 or epoch == (num_epochs - 1):
            save_checkpoint(epoch + 1, latest_path)
            if cfg.save_every_freq > 0 and epoch % cfg.save_every_freq == 0:
                save_checkpoint(epoch + 1, os.path.join(folder, f'{tag}-e{epoch}.pth.tar'))
