"""The reference-style entry point end to end on the GPU: `main(args)` with the reference's YAML schema on a tiny
synthetic config -> CSV log with the reference's columns, decreasing loss, checkpoint with the reference's keys
(`module.backbone.*`), and resume (schedules / mask counter replayed, train.py:322-326)."""
import csv
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


def tiny_args(folder, epochs):
    mask = dict(aspect_ratio=[0.75, 1.5], num_blocks=2, spatial_scale=[0.15, 0.15], temporal_scale=[1.0, 1.0],
                max_temporal_keep=1.0, max_keep=None)
    return {
        'meta': {'load_checkpoint': False, 'read_checkpoint': None, 'seed': 234, 'use_sdpa': True, 'dtype': 'bfloat16'},
        'mask': [mask, dict(mask, num_blocks=1, spatial_scale=[0.5, 0.5])],
        'model': {'model_name': 'vit_tiny', 'pred_depth': 2, 'pred_embed_dim': 96, 'uniform_power': True,
                  'use_mask_tokens': True, 'zero_init_mask_tokens': True},
        'data': {'dataset_type': 'synthetic', 'datasets': [], 'batch_size': 4, 'num_clips': 1, 'num_frames': 8,
                 'tubelet_size': 2, 'sampling_rate': 4, 'crop_size': 64, 'patch_size': 16, 'pin_mem': False,
                 'num_workers': 0},
        'data_aug': {},
        'loss': {'loss_exp': 1.0, 'reg_coeff': 0.0},
        'optimization': {'ipe': 6, 'ipe_scale': 1.25, 'clip_grad': 10.0, 'weight_decay': 0.04,
                         'final_weight_decay': 0.4, 'epochs': epochs, 'warmup': 1, 'start_lr': 2e-4, 'lr': 6.25e-4,
                         'final_lr': 1e-6, 'ema': [0.998, 1.0]},
        'logging': {'folder': folder, 'write_tag': 'jepa'},
    }


def read_csv(path):
    with open(path) as f:
        return list(csv.DictReader(f))


def test_main_trains_logs_checkpoints_and_resumes(tmp_path):
    from jepa_amd.app.vjepa.train import main
    folder = str(tmp_path)
    main(tiny_args(folder, epochs=3))
    rows = read_csv(os.path.join(folder, 'jepa_r0.csv'))
    assert list(rows[0].keys()) == ['epoch', 'itr', 'loss', 'loss-jepa', 'reg-loss', 'enc-grad-norm',
                                    'pred-grad-norm', 'gpu-time(ms)', 'wall-time(ms)']
    assert len(rows) == 18
    losses = [float(r['loss']) for r in rows]
    assert all(l == l and l > 0 for l in losses)
    assert sum(losses[-6:]) / 6 < sum(losses[:6]) / 6          # the latent loss goes down
    # epoch 3 (index 2) > warmup (1): clipping active, norms logged
    assert float(rows[-1]['enc-grad-norm']) > 0 and float(rows[0]['enc-grad-norm']) == 0
    ck = torch.load(os.path.join(folder, 'jepa-latest.pth.tar'), map_location='cpu')
    assert set(ck) >= {'encoder', 'predictor', 'opt', 'scaler', 'target_encoder', 'epoch', 'loss', 'batch_size',
                       'world_size', 'lr'}
    assert ck['epoch'] == 3
    assert 'module.backbone.blocks.0.attn.qkv.weight' in ck['encoder']
    assert 'module.backbone.predictor_blocks.1.mlp.fc2.bias' in ck['predictor']
    assert 'module.backbone.pos_embed' in ck['target_encoder']
    # like torch.optim.AdamW built by the reference's init_opt: the two frozen sincos tables (pos_embed, predictor_pos_embed)
    # are members of groups 0 / 1 and consume a parameter id, but carry no state
    n_state = sum(1 for _ in ck['opt']['state'])
    n_members = sum(len(g['params']) for g in ck['opt']['param_groups'])
    assert n_state == n_members - 2
    assert sorted(i for g in ck['opt']['param_groups'] for i in g['params']) == list(range(n_members))
    # EMA moved the target away from its initial copy but keeps it close to the encoder
    e = ck['encoder']['module.backbone.blocks.0.attn.qkv.weight']
    t = ck['target_encoder']['module.backbone.blocks.0.attn.qkv.weight']
    assert 0 < (e - t).abs().max() < 0.1
    # resume: picks up at epoch 3 and runs one more epoch
    main(tiny_args(folder, epochs=4), resume_preempt=True)
    rows2 = read_csv(os.path.join(folder, 'jepa_r0.csv'))
    tail = [r for r in rows2 if r['epoch'] == '4']
    assert len(tail) == 6
    assert abs(float(tail[0]['loss']) - losses[-1]) < 0.2 * losses[-1]   # continues from the trained state
