"""Sharp small-model parity: the HIP step against the oracle WITH bf16 storage emulation (oracle/vjepa_oracle.py, `emu=True`).

The fp32 oracle differs from the HIP path by the bf16 rounding of every stored activation / gradient: 1e-2 on features and up to
6e-2 ... 8e-2 on gradients of few-row tensors at the tiny sizes, which is what the bounds of tests/test_step_gpu.py have to allow
("a 3x kernel regression would pass at small sizes").  With the emulation the oracle rounds at the same points
(same functions, same arithmetic, only `.to(bfloat16)` round trips added; the fp32 path stays pinned by the golden fixtures), the
rounding noise becomes common-mode and the remaining difference is summation order plus the few roundings that cannot be
placed identically (the attention backward's recomputed probabilities, the online soft-max's per-tile rounding of P).  Bounds below
are 2x the values measured on the MI355X (printed by the tests); every one of them is several times tighter than the fp32 bound.
"""
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.golden_util import HP, MICRO, load_micro, micro_weights, rel_l2, step_inputs  # noqa: E402
from tests.step_util import TINY, TINY_MASKS, build_trainer, draw_batch, oracle_cfg, to_dev  # noqa: E402

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _compare(tr, o_emu, g_emu, o_f32, g_f32, out, tag, bounds):
    """-> dict of measured errors vs the emulating oracle (and, for context, vs the fp32 oracle)."""
    res = {}
    res["loss"] = abs(out.loss - o_emu["loss"]) / abs(o_emu["loss"])
    res["loss_f32"] = abs(out.loss - o_f32["loss"]) / abs(o_f32["loss"])
    worst, worst32, name_w = 0.0, 0.0, None
    per = []
    for grp in ("enc", "pred"):
        for n, ge in g_emu[grp].items():
            g = tr.arena.grad(grp + "." + n).float().cpu().reshape(ge.shape)
            e, e32 = rel_l2(g, ge), rel_l2(g, g_f32[grp][n])
            per.append((e, e32, grp + "." + n))
            if e > worst:
                worst, name_w = e, grp + "." + n
            worst32 = max(worst32, e32)
    res["grad_worst"], res["grad_worst_f32"], res["grad_worst_name"] = worst, worst32, name_w
    per.sort(reverse=True)
    med = per[len(per) // 2]
    print(f"[{tag}] loss rel. error vs emulating oracle {res['loss']:.2e} (vs fp32 oracle {res['loss_f32']:.2e}); gradients: worst "
          f"{worst:.2e} ({name_w}; vs fp32 oracle worst {worst32:.2e}), median {med[0]:.2e} (fp32: {med[1]:.2e}); "
          f"five worst: {[(round(a, 5), n) for a, _, n in per[:5]]}")
    assert res["loss"] < bounds["loss"], (tag, res)
    assert worst < bounds["grad"], (tag, res, per[:5])
    if "grad_rest" in bounds:   # every tensor except the named ones against a tighter bound
        rest = [x for x in per if x[2] not in bounds["except"]]
        assert rest[0][0] < bounds["grad_rest"], (tag, rest[:5])
    return res


def test_micro_step_vs_bf16_emulating_oracle():
    """The D = 64 fixture model (weights and inputs of tests/golden/micro_step.npz, produced by the REAL reference): forward
    features and every gradient of the HIP step against the emulating oracle."""
    import copy
    from oracle import vjepa_oracle as O
    from jepa_amd.engine.step import Trainer
    from tests.test_step_gpu import build_micro_modules, load_into
    z = load_micro()
    enc_w, pred_w = micro_weights(z)
    enc, pred = build_micro_modules()
    load_into(enc, enc_w)
    load_into(pred, pred_w)
    tgt = copy.deepcopy(enc)
    for p in tgt.parameters():
        p.requires_grad = False
    enc.to(DEV), pred.to(DEV), tgt.to(DEV)
    tr = Trainer(enc, pred, tgt, loss_exp=HP["loss_exp"], reg_coeff=HP["reg_coeff"], betas=HP["betas"], eps=HP["eps"], device=DEV)
    state = dict(enc=enc_w, pred=pred_w, tgt={k: v.clone() for k, v in enc_w.items()}, opt={})
    clips, me, mp = step_inputs(z, 0)
    o32, g32 = O.step_grads(state, clips, me, mp, dict(MICRO), HP)
    oe, ge = O.step_grads(state, clips, me, mp, dict(MICRO), HP, emu=True)
    cd, med, mpd = to_dev(clips, me, mp)
    h = tr.forward_target(cd, mpd)
    with torch.no_grad():
        zenc = enc(cd, med)
        zpred = pred(zenc, h, med, mpd)
    for i in range(2):
        eh, ez, ep = rel_l2(h[i].cpu(), oe["h"][i]), rel_l2(zenc[i].float().cpu(), oe["z_enc"][i]), rel_l2(zpred[i].float().cpu(), oe["z"][i])
        print(f"[micro] mask {i}: h {eh:.2e} (fp32 oracle {rel_l2(h[i].cpu(), o32['h'][i]):.2e}), z_enc {ez:.2e} "
              f"({rel_l2(zenc[i].float().cpu(), o32['z_enc'][i]):.2e}), z {ep:.2e} ({rel_l2(zpred[i].float().cpu(), o32['z'][i]):.2e})")
        assert eh < 1e-3 and ez < 1e-3 and ep < 2e-3, (i, eh, ez, ep)          # measured 2.1e-4 / 1.6e-4 / 7.3e-4; fp32 oracle: 5e-3
    out = tr.train_step(cd, med, mpd, lr=0.0, wd=0.0, ema=1.0)
    _compare(tr, oe, ge, o32, g32, out, "micro", dict(loss=2e-5, grad=1.5e-2))   # measured 3.1e-6 / 6.9e-3; fp32 oracle: 2.2e-5 / 2.6e-2


def _build_shallow(c, seed):
    """Encoder / predictor of arbitrary small dimensions through the reference-named module classes (as build_micro_modules)."""
    from functools import partial
    import torch.nn as nn
    from jepa_amd.src.models.predictor import VisionTransformerPredictor
    from jepa_amd.src.models.utils.multimask import MultiMaskWrapper, PredictorMultiMaskWrapper
    from jepa_amd.src.models.vision_transformer import VisionTransformer
    torch.manual_seed(seed)
    enc = VisionTransformer(img_size=c["crop"], patch_size=c["patch"], num_frames=c["frames"], tubelet_size=c["tubelet"],
                            embed_dim=c["embed_dim"], depth=c["depth"], num_heads=c["heads"], mlp_ratio=4, qkv_bias=True,
                            norm_layer=partial(nn.LayerNorm, eps=1e-6), uniform_power=True)
    pred = VisionTransformerPredictor(img_size=c["crop"], patch_size=c["patch"], num_frames=c["frames"], tubelet_size=c["tubelet"],
                                      embed_dim=c["embed_dim"], predictor_embed_dim=c["pred_dim"], depth=c["pred_depth"],
                                      num_heads=c["heads"], mlp_ratio=4, qkv_bias=True, norm_layer=partial(nn.LayerNorm, eps=1e-6),
                                      uniform_power=True, use_mask_tokens=True, num_mask_tokens=c["num_mask_tokens"],
                                      zero_init_mask_tokens=True)
    g = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():   # non-trivial biases / LayerNorm affines / mask tokens (0 / 1 at init)
        for mod in (enc, pred):
            for n, p in mod.named_parameters():
                if p.requires_grad and (p.dim() == 1 or "mask_tokens" in n):
                    p.add_(0.05 * torch.randn(p.shape, generator=g))
    return MultiMaskWrapper(enc), PredictorMultiMaskWrapper(pred)


SHALLOW = [
    # encoder head_dim 64 / predictor head_dim 32 (the ViT-L encoder's head size; HDP = 64 and 32 attention kernels)
    dict(embed_dim=192, depth=2, heads=3, pred_dim=96, pred_depth=2, num_mask_tokens=2, crop=64, frames=8, patch=16, tubelet=2,
         num_patches=64),
    # predictor head_dim 24 (the step's predictor: padded head, pad-column row sums), encoder head_dim 40 (HDP = 64, ragged)
    dict(embed_dim=160, depth=2, heads=4, pred_dim=96, pred_depth=2, num_mask_tokens=2, crop=64, frames=8, patch=16, tubelet=2,
         num_patches=64),
    # three blocks, head_dim 80 (ViT-H) / 128 in the predictor
    dict(embed_dim=160, depth=3, heads=2, pred_dim=256, pred_depth=1, num_mask_tokens=2, crop=64, frames=8, patch=16, tubelet=2,
         num_patches=64),
]


@pytest.mark.parametrize("ci", range(len(SHALLOW)))
def test_shallow_models_vs_bf16_emulating_oracle(ci):
    """Two- / three-block models of several widths and head sizes (every kernel family of the step: tubelet pack, GEMM epilogues,
    attention forward / backward per head-dim class, LayerNorm forward / backward with column sums, gather / scatter, predictor
    assembly, loss) against the emulating oracle.  At this depth the common-mode argument holds and the bounds are sharp."""
    import copy
    from oracle import vjepa_oracle as O
    from jepa_amd.engine.step import Trainer
    from tests.golden_util import MICRO_MASKS
    c = SHALLOW[ci]
    enc, pred = _build_shallow(c, 11 + ci)
    state = dict(enc={k[len("backbone."):]: v.detach().clone() for k, v in enc.state_dict().items()},
                 pred={k[len("backbone."):]: v.detach().clone() for k, v in pred.state_dict().items()}, opt={})
    state["tgt"] = {k: v.clone() for k, v in state["enc"].items()}
    tgt = copy.deepcopy(enc)
    for p in tgt.parameters():
        p.requires_grad = False
    enc.to(DEV), pred.to(DEV), tgt.to(DEV)
    tr = Trainer(enc, pred, tgt, device=DEV)
    gens = O.make_mask_gens(MICRO_MASKS, c["crop"], c["frames"], c["patch"], c["tubelet"])
    clips, me, mp = draw_batch(gens, 3, c, 300 + ci, 400 + ci)
    hp = dict(HP, reg_coeff=0.0)
    cfg = {k: c[k] for k in ("embed_dim", "depth", "heads", "pred_dim", "pred_depth", "num_mask_tokens", "patch", "tubelet", "num_patches")}
    o32, g32 = O.step_grads(state, clips, me, mp, cfg, hp)
    oe, ge = O.step_grads(state, clips, me, mp, cfg, hp, emu=True)
    cd, med, mpd = to_dev(clips, me, mp)
    h = tr.forward_target(cd, mpd)
    with torch.no_grad():
        zenc = enc(cd, med)
        zpred = pred(zenc, h, med, mpd)
    for i in range(2):
        eh, ez, ep = rel_l2(h[i].cpu(), oe["h"][i]), rel_l2(zenc[i].float().cpu(), oe["z_enc"][i]), rel_l2(zpred[i].float().cpu(), oe["z"][i])
        print(f"[shallow {ci}] mask {i}: h {eh:.2e} (fp32 oracle {rel_l2(h[i].cpu(), o32['h'][i]):.2e}), z_enc {ez:.2e} "
              f"({rel_l2(zenc[i].float().cpu(), o32['z_enc'][i]):.2e}), z {ep:.2e} ({rel_l2(zpred[i].float().cpu(), o32['z'][i]):.2e})")
        # measured (trip 15): h <= 6.8e-4, z_enc <= 9.0e-4, z <= 2.9e-3; against the fp32 oracle the same outputs sit at 5e-3 ... 6.6e-3
        assert eh < 2e-3 and ez < 2e-3 and ep < 6e-3, (ci, i, eh, ez, ep)
    out = tr.train_step(cd, med, mpd, lr=0.0, wd=0.0, ema=1.0)
    # measured: loss 5e-6 ... 9e-6, worst gradient 1.2e-2 (patch_embed.proj.weight, the end of the backward chain), median 6e-3;
    # fp32 oracle: worst 2.4e-2 ... 2.7e-2, median 1.6e-2
    _compare(tr, oe, ge, o32, g32, out, f"shallow {ci}", dict(loss=5e-5, grad=2.5e-2))


@pytest.mark.parametrize("B", [2, 5])
def test_tiny_step_vs_bf16_emulating_oracle(B):
    """ViT-Tiny 8x64x64 (BASELINE configs[0], TWELVE blocks), perturbed biases / LayerNorm affines / mask tokens.  At this depth the
    roundings de-correlate (a value that differs by 1e-4 relative flips its bf16 rounding with probability 1e-4 / 2^-8, and the
    attention kernels' per-tile rounding of P and the backward's recomputed P cannot be placed identically), so the emulating
    oracle is only modestly closer than the fp32 one (measured worst gradient 3.7e-2 vs 4.2e-2): the bound is the fp32 bound."""
    from oracle import vjepa_oracle as O
    tr, state, enc, pred, tgt = build_trainer(TINY, len(TINY_MASKS), perturb_small=True)
    gens = O.make_mask_gens(TINY_MASKS, TINY["crop"], TINY["frames"], TINY["patch"], TINY["tubelet"])
    clips, me, mp = draw_batch(gens, B, TINY, 77 + B, 99 + B)
    hp = dict(HP, reg_coeff=0.0)
    cfg = oracle_cfg(TINY, len(TINY_MASKS))
    o32, g32 = O.step_grads(state, clips, me, mp, cfg, hp)
    oe, ge = O.step_grads(state, clips, me, mp, cfg, hp, emu=True)
    cd, med, mpd = to_dev(clips, me, mp)
    out = tr.train_step(cd, med, mpd, lr=0.0, wd=0.0, ema=1.0)
    _compare(tr, oe, ge, o32, g32, out, f"tiny B={B}", dict(loss=2e-4, grad=6e-2))


@pytest.mark.timeout(900)
def test_vitl_step_vs_emulating_oracle_run_by_eager_pytorch_on_the_gpu():
    """Context figure at the benched MODEL (ViT-L/16 16x224x224, 24 + 12 blocks; B = 8 because the emulation materialises the
    [B, H, S, S] probabilities that SDPA never stores): the emulating oracle and the fp32 oracle, both run by eager PyTorch on the
    same GPU, against the HIP step.  With thousands of rows per tensor the de-correlated part of the rounding noise averages out and the
    emulation helps again: measured (trip 17) loss 2.9e-7 (fp32 oracle 3.4e-5), gradients median 2.0e-3 (7.8e-3), every tensor <= 3.4e-3
    except patch_embed.proj.weight, the end of the backward chain, at 1.1e-2 (1.3e-2)."""
    from oracle import vjepa_oracle as O
    from tests.step_util import VITL, VITL_MASKS
    tr, state, _, _, _ = build_trainer(VITL, 2)
    gens = O.make_mask_gens(VITL_MASKS, VITL["crop"], VITL["frames"], VITL["patch"], VITL["tubelet"])
    clips, me, mp = draw_batch(gens, 8, VITL, 1234, 4321)
    cd, med, mpd = to_dev(clips, me, mp)
    cfg = oracle_cfg(VITL, 2)
    st = {k: ({n: t.to(DEV) for n, t in v.items()} if k != "opt" else {}) for k, v in state.items()}
    hp = dict(HP, reg_coeff=0.0)
    o32, g32 = O.step_grads(st, cd, med, mpd, cfg, hp)
    g32 = {grp: {n: t.float().cpu() for n, t in gs.items()} for grp, gs in g32.items()}
    torch.cuda.empty_cache()
    oe, ge = O.step_grads(st, cd, med, mpd, cfg, hp, emu=True)
    ge = {grp: {n: t.float().cpu() for n, t in gs.items()} for grp, gs in ge.items()}
    del st
    torch.cuda.empty_cache()
    out = tr.train_step(cd, med, mpd, lr=0.0, wd=0.0, ema=1.0)
    _compare(tr, oe, ge, o32, g32, out, "ViT-L B=8", dict(loss=2e-5, grad=2.5e-2, grad_rest=7e-3, **{"except": ("enc.patch_embed.proj.weight",)}))
