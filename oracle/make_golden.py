"""Generate tests/golden/*.npz from the REAL reference (facebookresearch/jepa, mounted read-only at
/root/reference).  Run in the build container only:  python oracle/make_golden.py

The reference modules are imported unmodified; only the nested closure app/vjepa/train.py:414-498 (not importable:
it is a local function, and app.vjepa.train itself needs torchvision) is driven from here, through the
reference's own modules, optimizer factory (app.vjepa.utils.init_opt -> torch.optim.AdamW + schedulers) and
collator.  Outputs are the pins the oracle restatement and the HIP path are tested against.

Fixtures:
  micro_step.npz  -- "micro" V-JEPA (D=64, depth 2, 2 heads; predictor D=32, depth 2; 8x64x64 clips, B=2,
                     2 masks): weights, clips, mask indices and, for 2 consecutive steps, targets h, context
                     features, predictions z, losses, lr/wd/ema, selected gradients and post-step weights.
  host_tables.npz -- sincos position tables, collator draws, schedule values for the ViT-L recipe.
"""
import copy
import os
import sys

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")

MICRO = dict(embed_dim=64, depth=2, heads=2, pred_dim=32, pred_depth=2, num_mask_tokens=2, crop=64, frames=8,
             patch=16, tubelet=2)
MICRO_MASKS = [
    dict(aspect_ratio=(0.75, 1.5), num_blocks=2, spatial_scale=(0.15, 0.15), temporal_scale=(1.0, 1.0),
         max_temporal_keep=1.0, max_keep=None),
    dict(aspect_ratio=(0.75, 1.5), num_blocks=1, spatial_scale=(0.5, 0.5), temporal_scale=(0.5, 1.0),
         max_temporal_keep=1.0, max_keep=None),
]
HP = dict(loss_exp=1.0, reg_coeff=0.0, ipe=10, ipe_scale=1.25, epochs=4, warmup=1, start_lr=2e-4, lr=6.25e-4,
          final_lr=1e-6, wd=0.04, final_wd=0.4, ema=(0.998, 1.0), betas=(0.9, 0.999), eps=1e-8)


def build_micro():
    from functools import partial
    import torch.nn as nn
    from src.models.vision_transformer import VisionTransformer
    from src.models.predictor import VisionTransformerPredictor
    from src.models.utils.multimask import MultiMaskWrapper, PredictorMultiMaskWrapper
    c = MICRO
    torch.manual_seed(0)
    enc = VisionTransformer(img_size=c["crop"], patch_size=c["patch"], num_frames=c["frames"],
                            tubelet_size=c["tubelet"], embed_dim=c["embed_dim"], depth=c["depth"],
                            num_heads=c["heads"], mlp_ratio=4, qkv_bias=True,
                            norm_layer=partial(nn.LayerNorm, eps=1e-6), uniform_power=True)
    pred = VisionTransformerPredictor(img_size=c["crop"], patch_size=c["patch"], num_frames=c["frames"],
                                      tubelet_size=c["tubelet"], embed_dim=c["embed_dim"],
                                      predictor_embed_dim=c["pred_dim"], depth=c["pred_depth"],
                                      num_heads=c["heads"], mlp_ratio=4, qkv_bias=True,
                                      norm_layer=partial(nn.LayerNorm, eps=1e-6), uniform_power=True,
                                      use_mask_tokens=True, num_mask_tokens=c["num_mask_tokens"],
                                      zero_init_mask_tokens=True)
    # make every bias / LayerNorm affine / mask token non-trivial so all gradient paths are exercised
    g = torch.Generator().manual_seed(123)
    with torch.no_grad():
        for m in (enc, pred):
            for n, p in m.named_parameters():
                if p.requires_grad and (p.dim() == 1 or "mask_tokens" in n):
                    p.add_(0.05 * torch.randn(p.shape, generator=g))
    return MultiMaskWrapper(enc), PredictorMultiMaskWrapper(pred)


def flat(sd, strip="backbone."):
    return {k[len(strip):] if k.startswith(strip) else k: v.detach().clone() for k, v in sd.items()}


def main():
    sys.path.insert(0, REF)
    import torch.nn.functional as F
    from app.vjepa.utils import init_opt
    from src.masks.multiblock3d import MaskCollator
    from src.masks.utils import apply_masks
    from src.models.utils.pos_embs import get_3d_sincos_pos_embed
    from src.utils.schedulers import WarmupCosineSchedule, CosineWDSchedule
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(8)

    encoder, predictor = build_micro()
    target = copy.deepcopy(encoder)
    for p in target.parameters():
        p.requires_grad = False
    optimizer, _, scheduler, wd_scheduler = init_opt(
        encoder=encoder, predictor=predictor, wd=HP["wd"], final_wd=HP["final_wd"], start_lr=HP["start_lr"],
        ref_lr=HP["lr"], final_lr=HP["final_lr"], iterations_per_epoch=HP["ipe"], warmup=HP["warmup"],
        num_epochs=HP["epochs"], ipe_scale=HP["ipe_scale"], mixed_precision=False, betas=HP["betas"], eps=HP["eps"])
    momentum = (HP["ema"][0] + i * (HP["ema"][1] - HP["ema"][0]) / (HP["ipe"] * HP["epochs"] * HP["ipe_scale"])
                for i in range(int(HP["ipe"] * HP["epochs"] * HP["ipe_scale"]) + 1))
    collator = MaskCollator(cfgs_mask=MICRO_MASKS, crop_size=MICRO["crop"], num_frames=MICRO["frames"],
                            patch_size=MICRO["patch"], tubelet_size=MICRO["tubelet"])

    out = {}
    for k, v in flat(encoder.state_dict()).items():
        out["w0/enc/" + k] = v.numpy()
    for k, v in flat(predictor.state_dict()).items():
        out["w0/pred/" + k] = v.numpy()
    B = 2
    watch_enc = ["blocks.0.attn.qkv.weight", "blocks.1.mlp.fc2.bias", "patch_embed.proj.weight", "norm.weight",
                 "blocks.0.norm1.bias", "blocks.1.attn.proj.weight", "patch_embed.proj.bias"]
    watch_pred = ["mask_tokens.0", "mask_tokens.1", "predictor_embed.weight", "predictor_proj.bias",
                  "predictor_blocks.1.mlp.fc1.weight", "predictor_norm.bias", "predictor_blocks.0.attn.qkv.bias"]
    for step in range(2):
        clips = torch.randn(B, 3, MICRO["frames"], MICRO["crop"], MICRO["crop"],
                            generator=torch.Generator().manual_seed(1234 + step))
        torch.manual_seed(4321 + step)
        batch = [([clips[b]], 0, [torch.arange(MICRO["frames"])]) for b in range(B)]
        udata, masks_enc, masks_pred = collator(batch)
        c = torch.cat(udata[0], dim=0)
        assert torch.equal(c, clips)
        # ---- train.py:414-487, through the reference modules ----
        new_lr, new_wd = scheduler.step(), wd_scheduler.step()
        with torch.no_grad():
            h = target(c)
            h = F.layer_norm(h, (h.size(-1),))
            h = apply_masks(h, masks_pred, concat=False)
        z_enc = encoder(c, masks_enc)
        z = predictor(z_enc, h, masks_enc, masks_pred)
        loss_jepa = 0.
        for zi, hi in zip(z, h):
            loss_jepa += torch.mean(torch.abs(zi - hi) ** HP["loss_exp"]) / HP["loss_exp"]
        loss_jepa /= len(masks_pred)
        pstd = sum([torch.sqrt(zi.var(dim=1) + 0.0001) for zi in z]) / len(z)
        loss_reg = torch.mean(F.relu(1. - pstd))
        loss = loss_jepa + HP["reg_coeff"] * loss_reg
        loss.backward()
        ge, gp = flat({n: p.grad for n, p in encoder.named_parameters() if p.grad is not None}), \
            flat({n: p.grad for n, p in predictor.named_parameters() if p.grad is not None})
        optimizer.step()
        optimizer.zero_grad()
        m = next(momentum)
        with torch.no_grad():
            for pq, pk in zip(encoder.parameters(), target.parameters()):
                pk.data.mul_(m).add_((1. - m) * pq.detach().data)
        pre = f"s{step}/"
        out[pre + "clips"] = clips.numpy()
        for i in range(len(masks_enc)):
            out[pre + f"masks_enc{i}"] = masks_enc[i].numpy()
            out[pre + f"masks_pred{i}"] = masks_pred[i].numpy()
            out[pre + f"h{i}"] = h[i].numpy()
            out[pre + f"z_enc{i}"] = z_enc[i].detach().numpy()
            out[pre + f"z{i}"] = z[i].detach().numpy()
        out[pre + "scalars"] = np.array([float(loss), float(loss_jepa), float(loss_reg), new_lr, new_wd, m],
                                        dtype=np.float64)
        we, wp, wt = flat(encoder.state_dict()), flat(predictor.state_dict()), flat(target.state_dict())
        for n in watch_enc:
            out[pre + "grad/enc/" + n] = ge[n].numpy()
            out[pre + "post/enc/" + n] = we[n].numpy()
            out[pre + "post/tgt/" + n] = wt[n].numpy()
        for n in watch_pred:
            out[pre + "grad/pred/" + n] = gp[n].numpy()
            out[pre + "post/pred/" + n] = wp[n].numpy()
        print(f"step {step}: loss {float(loss):.7f} lr {new_lr:.3e} wd {new_wd:.4f} ema {m:.6f} "
              f"Ke {[x.shape[1] for x in masks_enc]} Kp {[x.shape[1] for x in masks_pred]}")
    np.savez_compressed(os.path.join(OUT, "micro_step.npz"), **out)

    # ---- host tables: pos-embeds, collator draws, schedules (ViT-L recipe, configs/pretrain/vitl16.yaml) ----
    tabs = {}
    # small grids in full; the big production tables as every 53rd row + float64 checksums (keeps the fixture small)
    for D, gs, gd in [(192, 4, 4), (96, 4, 4), (64, 4, 4), (32, 4, 4)]:
        tabs[f"pos3d_{D}_{gs}_{gd}"] = get_3d_sincos_pos_embed(D, gs, gd, cls_token=False, uniform_power=True)
    for D, gs, gd, up in [(1024, 14, 8, True), (384, 14, 8, True), (1280, 24, 8, True), (768, 14, 8, False)]:
        t = get_3d_sincos_pos_embed(D, gs, gd, cls_token=False, uniform_power=up)
        tabs[f"pos3d_rows53_{D}_{gs}_{gd}_{int(up)}"] = t[::53]
        tabs[f"pos3d_sums_{D}_{gs}_{gd}_{int(up)}"] = np.array([t.sum(), (t * t).sum(), np.abs(t).sum()])
    vitl_masks = [
        dict(aspect_ratio=(0.75, 1.5), num_blocks=8, spatial_scale=(0.15, 0.15), temporal_scale=(1.0, 1.0),
             max_temporal_keep=1.0, max_keep=None),
        dict(aspect_ratio=(0.75, 1.5), num_blocks=2, spatial_scale=(0.7, 0.7), temporal_scale=(1.0, 1.0),
             max_temporal_keep=1.0, max_keep=None),
    ]
    coll = MaskCollator(cfgs_mask=vitl_masks, crop_size=224, num_frames=16, patch_size=16, tubelet_size=2)
    for it in range(3):
        torch.manual_seed(4321 + it)
        _, me, mp = coll([(torch.zeros(1), 0) for _ in range(6)])
        for i in range(2):
            tabs[f"vitl_it{it}_enc{i}"] = me[i].numpy()
            tabs[f"vitl_it{it}_pred{i}"] = mp[i].numpy()
    # max_keep + temporal-keep variants
    odd = [dict(aspect_ratio=(0.3, 3.0), num_blocks=3, spatial_scale=(0.2, 0.8), temporal_scale=(0.25, 1.0),
                max_temporal_keep=0.5, max_keep=100)]
    coll2 = MaskCollator(cfgs_mask=odd, crop_size=224, num_frames=16, patch_size=16, tubelet_size=2)
    for it in range(3):
        torch.manual_seed(99 + it)
        _, me, mp = coll2([(torch.zeros(1), 0) for _ in range(4)])
        tabs[f"odd_it{it}_enc0"] = me[0].numpy()
        tabs[f"odd_it{it}_pred0"] = mp[0].numpy()

    class _Opt:
        param_groups = [{"lr": 0., "weight_decay": 0.}, {"lr": 0., "weight_decay": 0., "WD_exclude": True}]
    o = _Opt()
    s = WarmupCosineSchedule(o, warmup_steps=int(40 * 300), start_lr=2e-4, ref_lr=6.25e-4, final_lr=1e-6,
                             T_max=int(1.25 * 300 * 300))
    ws = CosineWDSchedule(o, ref_wd=0.04, final_wd=0.4, T_max=int(1.25 * 300 * 300))
    probe = [1, 2, 100, 11999, 12000, 12001, 50000, 90000, 112500, 112600]
    lrs, wds, k = [], [], 0
    for step in range(1, probe[-1] + 1):
        a, b = s.step(), ws.step()
        if step == probe[k]:
            lrs.append(a)
            wds.append(b)
            k += 1
    tabs["sched_steps"] = np.array(probe)
    tabs["sched_lr"] = np.array(lrs)
    tabs["sched_wd"] = np.array(wds)
    np.savez_compressed(os.path.join(OUT, "host_tables.npz"), **tabs)
    for f in ("micro_step.npz", "host_tables.npz"):
        print(f, os.path.getsize(os.path.join(OUT, f)) // 1024, "KiB")


if __name__ == "__main__":
    main()
