"""Shared helpers for the parity tests: load the golden fixtures generated from the real reference
(oracle/make_golden.py) and build oracle / HIP-path states from them."""
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

MICRO = dict(embed_dim=64, depth=2, heads=2, pred_dim=32, pred_depth=2, num_mask_tokens=2, crop=64, frames=8,
             patch=16, tubelet=2, num_patches=4 * 4 * 4)
MICRO_MASKS = [
    dict(aspect_ratio=(0.75, 1.5), num_blocks=2, spatial_scale=(0.15, 0.15), temporal_scale=(1.0, 1.0),
         max_temporal_keep=1.0, max_keep=None),
    dict(aspect_ratio=(0.75, 1.5), num_blocks=1, spatial_scale=(0.5, 0.5), temporal_scale=(0.5, 1.0),
         max_temporal_keep=1.0, max_keep=None),
]
HP = dict(loss_exp=1.0, reg_coeff=0.0, ipe=10, ipe_scale=1.25, epochs=4, warmup=1, start_lr=2e-4, lr=6.25e-4,
          final_lr=1e-6, wd=0.04, final_wd=0.4, ema=(0.998, 1.0), betas=(0.9, 0.999), eps=1e-8)


def load_micro():
    return np.load(os.path.join(GOLDEN, "micro_step.npz"))


def load_tables():
    return np.load(os.path.join(GOLDEN, "host_tables.npz"))


def micro_weights(z):
    enc = {k[len("w0/enc/"):]: torch.from_numpy(z[k]).clone() for k in z.files if k.startswith("w0/enc/")}
    pred = {k[len("w0/pred/"):]: torch.from_numpy(z[k]).clone() for k in z.files if k.startswith("w0/pred/")}
    return enc, pred


def step_inputs(z, s, n_masks=2):
    clips = torch.from_numpy(z[f"s{s}/clips"])
    me = [torch.from_numpy(z[f"s{s}/masks_enc{i}"]) for i in range(n_masks)]
    mp = [torch.from_numpy(z[f"s{s}/masks_pred{i}"]) for i in range(n_masks)]
    return clips, me, mp


def rel_l2(a, b):
    a, b = torch.as_tensor(a).double(), torch.as_tensor(b).double()
    return float((a - b).norm() / (b.norm() + 1e-300))
