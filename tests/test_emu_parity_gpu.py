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
        assert eh < 4e-3 and ez < 4e-3 and ep < 4e-3, (i, eh, ez, ep)
    out = tr.train_step(cd, med, mpd, lr=0.0, wd=0.0, ema=1.0)
    _compare(tr, oe, ge, o32, g32, out, "micro", dict(loss=2e-4, grad=2e-2))


@pytest.mark.parametrize("B", [2, 5])
def test_tiny_step_vs_bf16_emulating_oracle(B):
    """ViT-Tiny 8x64x64 (BASELINE configs[0]), perturbed biases / LayerNorm affines / mask tokens so that no gradient is trivially
    zero: loss and every gradient of the arena against the emulating oracle."""
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
    _compare(tr, oe, ge, o32, g32, out, f"tiny B={B}", dict(loss=2e-4, grad=2e-2))
