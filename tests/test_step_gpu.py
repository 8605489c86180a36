"""End-to-end parity of the HIP pretraining step against (a) the golden fixture produced by the REAL reference
(tests/golden/micro_step.npz) and (b) the pinned oracle on the ViT-Tiny plumbing config (BASELINE configs[0]).

Stated tolerances (bf16 compute vs fp32 reference; bf16 eps = 2^-8 ~ 3.9e-3):
  targets h / context features / predictions : rel-L2 <= 2e-2
  loss                                        : <= 1e-3 relative  (north-star bound)
  gradients                                   : rel-L2 <= 6e-2, cosine >= 0.998
  AdamW/EMA-updated weights                   : within 2.5 * lr per element of the reference (first steps of Adam
                                                move every weight by ~lr * sign(g); a flipped sign on a
                                                near-zero gradient costs at most 2*lr), rel-L2 <= 2e-3
"""
import os
import sys
from functools import partial

import pytest
import torch
import torch.nn as nn

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.golden_util import HP, MICRO, load_micro, micro_weights, rel_l2, step_inputs  # noqa: E402

pytestmark = pytest.mark.gpu
DEV = "cuda"


def build_micro_modules():
    from jepa_amd.src.models.predictor import VisionTransformerPredictor
    from jepa_amd.src.models.utils.multimask import MultiMaskWrapper, PredictorMultiMaskWrapper
    from jepa_amd.src.models.vision_transformer import VisionTransformer
    c = MICRO
    enc = VisionTransformer(img_size=c["crop"], patch_size=c["patch"], num_frames=c["frames"],
                            tubelet_size=c["tubelet"], embed_dim=c["embed_dim"], depth=c["depth"],
                            num_heads=c["heads"], mlp_ratio=4, qkv_bias=True,
                            norm_layer=partial(nn.LayerNorm, eps=1e-6), uniform_power=True)
    pred = VisionTransformerPredictor(img_size=c["crop"], patch_size=c["patch"], num_frames=c["frames"],
                                      tubelet_size=c["tubelet"], embed_dim=c["embed_dim"],
                                      predictor_embed_dim=c["pred_dim"], depth=c["pred_depth"], num_heads=c["heads"],
                                      mlp_ratio=4, qkv_bias=True, norm_layer=partial(nn.LayerNorm, eps=1e-6),
                                      uniform_power=True, use_mask_tokens=True,
                                      num_mask_tokens=c["num_mask_tokens"], zero_init_mask_tokens=True)
    return MultiMaskWrapper(enc), PredictorMultiMaskWrapper(pred)


def load_into(wrapper, weights):
    sd = {"backbone." + k: v for k, v in weights.items()}
    missing = wrapper.load_state_dict(sd, strict=True)
    return missing


def cosine(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float((a @ b) / (a.norm() * b.norm() + 1e-300))


@pytest.fixture(scope="module")
def micro_trainer():
    import copy
    from jepa_amd.engine.step import Trainer
    z = load_micro()
    enc_w, pred_w = micro_weights(z)
    enc, pred = build_micro_modules()
    load_into(enc, enc_w)
    load_into(pred, pred_w)
    tgt = copy.deepcopy(enc)
    for p in tgt.parameters():
        p.requires_grad = False
    enc.to(DEV), pred.to(DEV), tgt.to(DEV)
    tr = Trainer(enc, pred, tgt, loss_exp=HP["loss_exp"], reg_coeff=HP["reg_coeff"], betas=HP["betas"],
                 eps=HP["eps"], device=DEV)
    return z, tr, enc, pred, tgt


def test_state_dict_names_match_reference(micro_trainer):
    z, tr, enc, pred, tgt = micro_trainer
    enc_keys = {k[len("w0/enc/"):] for k in z.files if k.startswith("w0/enc/")}
    pred_keys = {k[len("w0/pred/"):] for k in z.files if k.startswith("w0/pred/")}
    assert {k[len("backbone."):] for k in enc.state_dict()} == enc_keys
    assert {k[len("backbone."):] for k in pred.state_dict()} == pred_keys


def test_micro_two_steps_vs_reference_fixture(micro_trainer):
    from oracle import vjepa_oracle as O
    z, tr, enc, pred, tgt = micro_trainer
    for s in range(2):
        clips, me, mp = step_inputs(z, s)
        clips_d, me_d, mp_d = clips.to(DEV), [m.to(DEV) for m in me], [m.to(DEV) for m in mp]
        sc = z[f"s{s}/scalars"]
        lr, wd, ema = float(sc[3]), float(sc[4]), float(sc[5])
        # --- forward pieces through the public module API (inference path), before the update
        h = tr.forward_target(clips_d, mp_d)
        with torch.no_grad():
            zenc = enc(clips_d, me_d)
            zpred = pred(zenc, h, me_d, mp_d)
        for i in range(2):
            assert rel_l2(h[i].cpu(), z[f"s{s}/h{i}"]) < 2e-2, ("h", s, i, rel_l2(h[i].cpu(), z[f"s{s}/h{i}"]))
            assert rel_l2(zenc[i].float().cpu(), z[f"s{s}/z_enc{i}"]) < 2e-2, ("z_enc", s, i)
            assert rel_l2(zpred[i].float().cpu(), z[f"s{s}/z{i}"]) < 2e-2, ("z", s, i, rel_l2(zpred[i].float().cpu(), z[f"s{s}/z{i}"]))
        # --- the fused step
        out = tr.train_step(clips_d, me_d, mp_d, lr=lr, wd=wd, ema=ema)
        assert abs(out.loss_jepa - sc[1]) < 1e-3 * abs(sc[1]), (out.loss_jepa, sc[1])
        assert abs(out.loss - sc[0]) < 1e-3 * abs(sc[0]), (out.loss, sc[0])
        assert abs(out.loss_reg - sc[2]) < 2e-2 * abs(sc[2]) + 1e-4, (out.loss_reg, sc[2])
        for k in z.files:
            if k.startswith(f"s{s}/grad/"):
                _, _, grp, name = k.split("/", 3)
                g = tr.arena.grad(("enc." if grp == "enc" else "pred.") + name).float().cpu().reshape(z[k].shape)
                ref = torch.from_numpy(z[k])
                assert rel_l2(g, ref) < 6e-2, (k, rel_l2(g, ref))
                assert cosine(g, ref) > 0.998, (k, cosine(g, ref))
            if k.startswith(f"s{s}/post/"):
                _, _, grp, name = k.split("/", 3)
                if grp == "tgt":
                    w = tr.tarena.f32("enc." + name)
                else:
                    w = tr.arena.f32(("enc." if grp == "enc" else "pred.") + name)
                w = w.float().cpu().reshape(z[k].shape)
                ref = torch.from_numpy(z[k])
                tol = 2.5 * lr * (1.0 if grp != "tgt" else (1 - ema) * 2)
                assert (w - ref).abs().max() <= tol + 1e-7, (k, float((w - ref).abs().max()), tol)
                assert rel_l2(w, ref) < 2e-3, (k, rel_l2(w, ref))


def test_gather_inside_step_is_bit_exact(micro_trainer):
    """apply_masks on the GPU == torch.gather on the same bits (north-star: mask gather bit-exact)."""
    from jepa_amd.src.masks.utils import apply_masks
    z, tr, enc, pred, tgt = micro_trainer
    clips, me, mp = step_inputs(z, 0)
    x = torch.randn(2, MICRO["num_patches"], MICRO["embed_dim"], device=DEV)
    out = apply_masks(x, [m.to(DEV) for m in mp], concat=False)
    for o, m in zip(out, mp):
        ref = torch.gather(x.cpu(), 1, m.unsqueeze(-1).repeat(1, 1, x.shape[-1]))
        assert torch.equal(o.cpu(), ref)


def test_module_autograd_path_matches_oracle():
    """encoder(clips, masks) -> predictor(...) -> loss.backward() through the module-level autograd nodes."""
    from oracle import vjepa_oracle as O
    z = load_micro()
    enc_w, pred_w = micro_weights(z)
    enc, pred = build_micro_modules()
    load_into(enc, enc_w)
    load_into(pred, pred_w)
    enc.to(DEV), pred.to(DEV)
    clips, me, mp = step_inputs(z, 0)
    me_d, mp_d = [m.to(DEV) for m in me], [m.to(DEV) for m in mp]
    zenc = enc(clips.to(DEV), me_d)
    zhat = pred(zenc, None, me_d, mp_d)
    h = [torch.from_numpy(z[f"s0/h{i}"]).to(DEV) for i in range(2)]
    loss = sum((a.float() - b).abs().mean() for a, b in zip(zhat, h)) / 2
    loss.backward()
    sc = z["s0/scalars"]
    assert abs(float(loss) - sc[1]) < 2e-3 * abs(sc[1])
    for k in z.files:
        if k.startswith("s0/grad/"):
            _, _, grp, name = k.split("/", 3)
            mod = enc if grp == "enc" else pred
            p = dict(mod.named_parameters())["backbone." + name]
            assert p.grad is not None, k
            assert rel_l2(p.grad.float().cpu(), z[k]) < 7e-2, (k, rel_l2(p.grad.float().cpu(), z[k]))


def test_vit_tiny_step_vs_oracle():
    """BASELINE configs[0]: ViT-Tiny/16, 8x64x64, B=2, 1 mask -- HIP step vs the pinned oracle on seeded inputs."""
    import copy
    from oracle import vjepa_oracle as O
    from jepa_amd.app.vjepa.utils import init_video_model
    from jepa_amd.engine.step import Trainer
    torch.manual_seed(0)
    enc, pred = init_video_model(device="cpu", patch_size=16, num_frames=8, tubelet_size=2, model_name="vit_tiny",
                                 crop_size=64, pred_depth=2, pred_embed_dim=96, uniform_power=True,
                                 use_mask_tokens=True, num_mask_tokens=1, zero_init_mask_tokens=True)
    with torch.no_grad():  # non-trivial biases / mask token
        g = torch.Generator().manual_seed(5)
        for m in (enc, pred):
            for n, p in m.named_parameters():
                if p.requires_grad and (p.dim() == 1 or "mask_tokens" in n):
                    p.add_(0.02 * torch.randn(p.shape, generator=g))
    cfg = dict(embed_dim=192, depth=12, heads=3, pred_dim=96, pred_depth=2, num_mask_tokens=1, patch=16, tubelet=2,
               num_patches=64)
    ow_enc = {k[len("backbone."):]: v.detach().clone() for k, v in enc.state_dict().items()}
    ow_pred = {k[len("backbone."):]: v.detach().clone() for k, v in pred.state_dict().items()}
    state = dict(enc=ow_enc, pred=ow_pred, tgt={k: v.clone() for k, v in ow_enc.items()}, opt={})
    tgt = copy.deepcopy(enc)
    for p in tgt.parameters():
        p.requires_grad = False
    enc.to(DEV), pred.to(DEV), tgt.to(DEV)
    tr = Trainer(enc, pred, tgt, device=DEV)
    masks_cfg = [dict(aspect_ratio=(0.75, 1.5), num_blocks=8, spatial_scale=(0.15, 0.15), temporal_scale=(1.0, 1.0))]
    gens = O.make_mask_gens(masks_cfg, 64, 8, 16, 2)
    hp = dict(HP)
    for step in range(1, 4):
        clips = torch.randn(2, 3, 8, 64, 64, generator=torch.Generator().manual_seed(1234 + step))
        torch.manual_seed(4321 + step)
        me, mp = zip(*[gq(2) for gq in gens])
        me, mp = list(me), list(mp)
        ref = O.train_step(state, clips, me, mp, cfg, hp, step)
        out = tr.train_step(clips.to(DEV), [m.to(DEV) for m in me], [m.to(DEV) for m in mp], lr=ref["lr"],
                            wd=ref["wd"], ema=ref["ema"])
        assert abs(out.loss - ref["loss"]) < 1e-3 * abs(ref["loss"]), (step, out.loss, ref["loss"])
        gq = tr.arena.grad("enc.blocks.11.attn.qkv.weight").float().cpu()
        # ViT-Tiny on 8x64x64 clips at B=2 keeps ~10-20 context tokens per sample: a weight gradient is a sum over so few
        # bf16-rounded rows that the rounding noise does not average out (measured 6.0e-2 here against 6e-3..1.4e-2 on the
        # same tensor class at ViT-L / ViT-H size, where the bound is 3e-2 for EVERY tensor: test_step_fullsize_gpu.py, arena-wide)
        assert rel_l2(gq, ref["grads"]["enc"]["blocks.11.attn.qkv.weight"]) < 8e-2
        gp = tr.arena.grad("enc.patch_embed.proj.weight").float().cpu()
        assert cosine(gp, ref["grads"]["enc"]["patch_embed.proj.weight"]) > 0.99
    w = tr.arena.f32("enc.blocks.0.mlp.fc1.weight").cpu()
    assert rel_l2(w, state["enc"]["blocks.0.mlp.fc1.weight"]) < 5e-3
    wt = tr.tarena.f32("enc.blocks.0.mlp.fc1.weight").cpu()
    assert rel_l2(wt, state["tgt"]["blocks.0.mlp.fc1.weight"]) < 1e-4


def test_variance_regulariser_backward_vs_oracle():
    """reg_coeff != 0 (train.py:448-459): loss and gradients of loss_jepa + reg_coeff * mean(relu(1 - pstd))."""
    import copy
    from oracle import vjepa_oracle as O
    from jepa_amd.engine.step import Trainer
    z = load_micro()
    enc_w, pred_w = micro_weights(z)
    # scale the predictor's output projection down so the token std is < 1 and the relu is active everywhere
    pred_w["predictor_proj.weight"] = pred_w["predictor_proj.weight"] * 0.25
    enc, pred = build_micro_modules()
    load_into(enc, enc_w)
    load_into(pred, pred_w)
    tgt = copy.deepcopy(enc)
    for p in tgt.parameters():
        p.requires_grad = False
    enc.to(DEV), pred.to(DEV), tgt.to(DEV)
    hp = dict(HP, reg_coeff=0.5)
    tr = Trainer(enc, pred, tgt, loss_exp=hp["loss_exp"], reg_coeff=hp["reg_coeff"], betas=hp["betas"], eps=hp["eps"],
                 device=DEV)
    clips, me, mp = step_inputs(z, 0)
    state = dict(enc={k: v.clone() for k, v in enc_w.items()}, pred={k: v.clone() for k, v in pred_w.items()},
                 tgt={k: v.clone() for k, v in enc_w.items()}, opt={})
    ref = O.train_step(state, clips, me, mp, MICRO, hp, 1)
    out = tr.train_step(clips.to(DEV), [m.to(DEV) for m in me], [m.to(DEV) for m in mp], lr=ref["lr"], wd=ref["wd"],
                        ema=ref["ema"])
    assert ref["loss_reg"] > 0.05, "test setup: the regulariser must be active"
    assert abs(out.loss_reg - ref["loss_reg"]) < 2e-2 * ref["loss_reg"], (out.loss_reg, ref["loss_reg"])
    assert abs(out.loss - ref["loss"]) < 2e-3 * abs(ref["loss"]), (out.loss, ref["loss"])
    for name in ("predictor_proj.weight", "predictor_blocks.1.mlp.fc1.weight", "mask_tokens.0"):
        g = tr.arena.grad("pred." + name).float().cpu().reshape(ref["grads"]["pred"][name].shape)
        # micro model (D = 64 / 32, ~30 tokens): measured 3.6e-2 on predictor_proj.weight and 7.3e-2 on mask_tokens.0 (a sum
        # over ~20 bf16 rows); 3e-2 is the bound at full size (every tensor: test_step_fullsize_gpu.py, arena-wide)
        assert rel_l2(g, ref["grads"]["pred"][name]) < 8e-2, (name, rel_l2(g, ref["grads"]["pred"][name]))
    g = tr.arena.grad("enc.blocks.0.attn.qkv.weight").float().cpu()
    assert cosine(g, ref["grads"]["enc"]["blocks.0.attn.qkv.weight"]) > 0.995


def _oracle_state(enc, pred):
    ow_enc = {k[len("backbone."):]: v.detach().clone() for k, v in enc.state_dict().items()}
    ow_pred = {k[len("backbone."):]: v.detach().clone() for k, v in pred.state_dict().items()}
    return dict(enc=ow_enc, pred=ow_pred, tgt={k: v.clone() for k, v in ow_enc.items()}, opt={})


def test_vit_tiny_loss_curve_20_steps_vs_oracle():
    """Per-step loss curve over 20 optimisation steps (same init, clips, masks, schedules): every step within
    1e-3 relative of the fp32 oracle, i.e. the bf16 HIP trajectory does not drift away from the reference one."""
    import copy
    from oracle import vjepa_oracle as O
    from jepa_amd.app.vjepa.utils import init_video_model
    from jepa_amd.engine.step import Trainer
    torch.manual_seed(0)
    enc, pred = init_video_model(device="cpu", patch_size=16, num_frames=8, tubelet_size=2, model_name="vit_tiny",
                                 crop_size=64, pred_depth=2, pred_embed_dim=96, uniform_power=True,
                                 use_mask_tokens=True, num_mask_tokens=2, zero_init_mask_tokens=True)
    cfg = dict(embed_dim=192, depth=12, heads=3, pred_dim=96, pred_depth=2, num_mask_tokens=2, patch=16, tubelet=2,
               num_patches=64)
    state = _oracle_state(enc, pred)
    tgt = copy.deepcopy(enc)
    for p in tgt.parameters():
        p.requires_grad = False
    enc.to(DEV), pred.to(DEV), tgt.to(DEV)
    tr = Trainer(enc, pred, tgt, device=DEV)
    masks_cfg = [dict(aspect_ratio=(0.75, 1.5), num_blocks=4, spatial_scale=(0.15, 0.15), temporal_scale=(1.0, 1.0)),
                 dict(aspect_ratio=(0.75, 1.5), num_blocks=1, spatial_scale=(0.6, 0.6), temporal_scale=(0.5, 1.0))]
    gens = O.make_mask_gens(masks_cfg, 64, 8, 16, 2)
    hp = dict(HP, ipe=20, warmup=0.25)
    worst = 0.0
    for step in range(1, 21):
        clips = torch.randn(4, 3, 8, 64, 64, generator=torch.Generator().manual_seed(77 + step))
        torch.manual_seed(999 + step)
        me, mp = zip(*[gq(4) for gq in gens])
        ref = O.train_step(state, clips, list(me), list(mp), cfg, hp, step)
        out = tr.train_step(clips.to(DEV), [m.to(DEV) for m in me], [m.to(DEV) for m in mp], lr=ref["lr"],
                            wd=ref["wd"], ema=ref["ema"])
        rel = abs(out.loss - ref["loss"]) / abs(ref["loss"])
        worst = max(worst, rel)
        assert rel < 1e-3, (step, out.loss, ref["loss"])
    print(f"worst relative loss deviation over 20 steps: {worst:.2e}")


def _big_model_step_vs_oracle(m, B, n_grad_checks=True):
    """One optimisation step of a BASELINE-shape model vs the fp32 CPU oracle on identical weights / clips / masks."""
    from oracle import vjepa_oracle as O
    from tests.step_util import VITL_MASKS, build_trainer, draw_batch, oracle_cfg, to_dev
    tr, state, enc, pred, tgt = build_trainer(m, 2)
    gens = O.make_mask_gens(VITL_MASKS, m["crop"], m["frames"], m["patch"], m["tubelet"])
    clips, me, mp = draw_batch(gens, B, m, 1234, 4321)
    torch.set_num_threads(min(os.cpu_count(), 64))
    ref = O.train_step(state, clips, me, mp, oracle_cfg(m, 2), dict(HP), 1)
    clips_d, me_d, mp_d = to_dev(clips, me, mp)
    report = {}
    if n_grad_checks:
        # forward pieces through the public module API (inference chains), before the update
        h = tr.forward_target(clips_d, mp_d)
        with torch.no_grad():
            zenc = enc(clips_d, me_d)
            zpred = pred(zenc, h, me_d, mp_d)
        for i in range(2):
            report[f"h{i}"] = rel_l2(h[i].cpu(), ref["h"][i])
            report[f"z_enc{i}"] = rel_l2(zenc[i].float().cpu(), ref["z_enc"][i])
            report[f"z{i}"] = rel_l2(zpred[i].float().cpu(), ref["z"][i])
    out = tr.train_step(clips_d, me_d, mp_d, lr=ref["lr"], wd=ref["wd"], ema=ref["ema"])
    report["loss_rel"] = abs(out.loss - ref["loss"]) / abs(ref["loss"])
    if n_grad_checks:
        L, Lp = m["depth"] - 1, m["pred_depth"] - 1
        for grp, name in (("enc", "patch_embed.proj.weight"), ("enc", "blocks.0.attn.qkv.weight"),
                          ("enc", f"blocks.{L}.mlp.fc2.weight"), ("enc", f"blocks.{L // 2}.attn.proj.weight"),
                          ("enc", f"blocks.{L}.mlp.fc2.bias"), ("enc", "norm.weight"),
                          ("pred", "predictor_blocks.0.attn.qkv.weight"), ("pred", f"predictor_blocks.{Lp}.mlp.fc1.weight"),
                          ("pred", "predictor_embed.weight"), ("pred", "mask_tokens.1")):
            g = tr.arena.grad(grp + "." + name).float().cpu()
            report[f"g:{grp}.{name}"] = rel_l2(g, ref["grads"][grp][name].reshape(g.shape))
    print(f"{m['model_name']} B={B}: HIP loss {out.loss:.6f} vs oracle {ref['loss']:.6f}; " +
          ", ".join(f"{k} {v:.2e}" for k, v in report.items()))
    return report


def test_vit_large_step_vs_oracle_baseline_shape():
    """BASELINE configs[1] model and clip shape (ViT-L/16, 16x224x224, vitl16.yaml masks), B=2, first step vs the fp32
    oracle: loss <= 1e-3 relative (north-star), targets / context features / predictions rel-L2 <= 2e-2, ten gradients
    (patch embed, first / middle / last encoder blocks, norms, predictor qkv / fc1 / embed, mask token) rel-L2 <= 3e-2."""
    from tests.step_util import VITL
    rep = _big_model_step_vs_oracle(VITL, 2)
    assert rep["loss_rel"] < 1e-3, rep
    for k, v in rep.items():
        if k[0] in "hz":
            assert v < 2e-2, (k, v)
        if k.startswith("g:"):
            assert v < 3e-2, (k, v)


def test_vit_huge_step_vs_oracle_head_dim_80():
    """BASELINE configs[3]/[4] model (ViT-H/16: 32 layers, D=1280, 16 heads -> head_dim 80, the native 96-wide attention
    class), 16x224x224, B=2: same bounds as the ViT-L test."""
    from tests.step_util import VITH
    rep = _big_model_step_vs_oracle(VITH, 2)
    assert rep["loss_rel"] < 1e-3, rep
    for k, v in rep.items():
        if k[0] in "hz":
            assert v < 2e-2, (k, v)
        if k.startswith("g:"):
            assert v < 3e-2, (k, v)


# (the benched batch, B = 24, is checked against the oracle run in fp32 by eager PyTorch on the GPU -- loss and EVERY gradient tensor -- in
#  tests/test_step_fullsize_gpu.py; the CPU-oracle loss check at that batch cost 80 - 200 s of host time and was dropped in round 6)
