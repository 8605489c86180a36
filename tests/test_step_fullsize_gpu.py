"""The step at BASELINE.json's sizes against the oracle run in fp32 by eager PyTorch on the same GPU (the CPU oracle needs minutes there):
ViT-L/16 B = 24 (every gradient tensor), ViT-H/16 16x384x384 (4608 tokens), a ten-step ViT-L loss curve, and size-independent
properties of the benched step (determinism, chains, micro-batches)."""
import os
import socket
import sys
import pytest
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.golden_util import HP, rel_l2  # noqa: E402
from tests.step_util import (TINY, TINY_MASKS, build_models, build_trainer, draw_batch, oracle_cfg,  # noqa: E402
                             to_dev)
from functools import partial
import torch.nn as nn
from tests.golden_util import HP, MICRO, load_micro, micro_weights, rel_l2, step_inputs  # noqa: E402
import math
from tests.golden_util import rel_l2  # noqa: E402
from tests.step_util import TINY, TINY_MASKS, build_trainer, draw_batch, to_dev  # noqa: E402
from tests.gpu_util import ATTN_SHAPES, bf, sdpa_ref  # noqa: E402,F401

pytestmark = pytest.mark.gpu


def rel_l2(a, b):
    a, b = a.float(), b.float()
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


DEV = "cuda"


def _arena_wide_gradient_check(tr, ref_grads, bound, what):
    """EVERY trainable tensor of encoder and predictor: rel-L2 of the HIP gradient (arena view) against the oracle's.
    Returns (worst value, its name); asserts every tensor under `bound` and prints the five worst."""
    errs = []
    for grp in ("enc", "pred"):
        for name, r in ref_grads[grp].items():
            g = tr.arena.grad(grp + "." + name).float()
            r = r.reshape(g.shape).float().to(g.device)
            errs.append((float((g - r).norm() / r.norm().clamp_min(1e-30)), grp + "." + name))
    errs.sort(reverse=True)
    print(f"{what}: {len(errs)} gradient tensors, worst rel-L2 " + ", ".join(f"{n} {e:.2e}" for e, n in errs[:5]))
    bad = [(n, e) for e, n in errs if not e < bound]
    assert not bad, (what, bad[:10])
    return errs[0]


# ------------------------------------------------------------------------------------------ the chain with / without bias_fuse
def fused_bias_mask(tr, arena=None):
    """bool mask over a parameter arena (default: the trainable one): True on the qkv / fc1 biases (the only gradients option
    bias_fuse changes)."""
    arena = tr.arena if arena is None else arena
    lo = getattr(arena, "lo", 0)                 # the EMA target arena covers the encoder range [lo, hi) of the trainer arena
    m = torch.zeros(arena.P.numel(), dtype=torch.bool, device=arena.P.device)
    for name, sl in arena.slots.items():
        if name.endswith("attn.qkv.bias") or name.endswith("mlp.fc1.bias"):
            m[sl.off - lo:sl.off - lo + sl.numel] = True
    return m


def cosine(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float((a @ b) / (a.norm() * b.norm() + 1e-300))


# ------------------------------------------------------------------------------------------------ BASELINE size
@pytest.mark.timeout(600)
def test_full_size_step_properties_vitl_b24():
    """Size-independent properties at the benched configuration (ViT-L/16, 16x224x224, B=24, vitl16.yaml masks), where
    the CPU oracle is too slow to run routinely: with lr = wd = 0 and ema = 1 a step leaves weights and target untouched,
    so the SAME step can be repeated under different execution modes and must reproduce
      * bit-identical losses and gradient arena: C launch chain vs per-kernel Python chain, and run-to-run (the split-K and
        partial reductions are deterministic);
      * the full-batch gradients from micro-batches of 12 and of 9 (uneven 9+9+6): rel-L2 <= 2e-5, loss <= 1e-6 relative;
      * finite gradients everywhere, step not skipped, unchanged weights."""
    from jepa_amd.engine import layers
    from tests.step_util import VITL, VITL_MASKS
    from oracle import vjepa_oracle as O
    tr, _, _, _, _ = build_trainer(VITL, 2)
    gens = O.make_mask_gens(VITL_MASKS, VITL["crop"], VITL["frames"], VITL["patch"], VITL["tubelet"])
    clips, me, mp = draw_batch(gens, 24, VITL, 1234, 4321)
    cd, med, mpd = to_dev(clips, me, mp)
    P0, T0 = tr.arena.P.clone(), tr.tarena.P.clone()

    def run(mb=None, c_chain=True):
        tr.micro_batch = mb
        layers.USE_C_CHAIN = c_chain
        try:
            o = tr.train_step(cd, med, mpd, lr=0.0, wd=0.0, ema=1.0)
            return o.loss, tr.arena.G.clone(), o.skipped
        finally:
            layers.USE_C_CHAIN = True
            tr.micro_batch = None
    l0, g0, sk = run()
    assert not sk and bool(torch.isfinite(g0).all()) and 0.1 < l0 < 5.0
    l1, g1, _ = run()
    assert l1 == l0 and torch.equal(g1, g0), "the step is not deterministic run-to-run"
    lp, gp, _ = run(c_chain=False)
    from tests.gpu_util import fused_bias_mask
    fm = fused_bias_mask(tr)     # (the qkv / fc1 bias gradients take the fused route in the C chain only, see test_chain_gpu.py)
    assert lp == l0 and torch.equal(gp[~fm], g0[~fm]), "C launch chain and Python chain diverge at full size"
    assert rel_l2(gp[fm].cpu(), g0[fm].cpu()) < 3e-3
    for mb in (12, 9):
        lm, gm, _ = run(mb=mb)
        assert abs(lm - l0) <= 1e-6 * abs(l0), (mb, lm, l0)
        r = float((gm.double() - g0.double()).norm() / g0.double().norm())
        assert r < 2e-5, (mb, r)
    assert torch.equal(tr.arena.P, P0) and torch.equal(tr.tarena.P, T0)


@pytest.mark.timeout(900)
def test_full_size_step_vs_the_oracle_run_by_eager_pytorch_on_the_gpu():
    """Parity at the benched batch in seconds instead of minutes: the oracle (the reference's arithmetic as plain torch
    functions) executed by stock PyTorch-ROCm eager on the SAME GPU -- fp32, and under autocast(bf16) as the reference runs on
    a GPU (train.py:419-438) -- against the HIP step on identical weights / clips / masks, ViT-L/16 16x224x224, B=24.
    Loss within 1e-3 relative of the fp32 run (north-star bound) and of the autocast run; every gradient tensor of the arena
    within 3e-2 rel-L2 of the fp32 run; prints the eager step time as the
    "reference on the same MI355X" context figure.  The oracle is the checker here, never the product."""
    import time
    from oracle import vjepa_oracle as O
    from tests.step_util import VITL, VITL_MASKS
    tr, state, _, _, _ = build_trainer(VITL, 2)
    gens = O.make_mask_gens(VITL_MASKS, VITL["crop"], VITL["frames"], VITL["patch"], VITL["tubelet"])
    clips, me, mp = draw_batch(gens, 24, VITL, 1234, 4321)
    cd, med, mpd = to_dev(clips, me, mp)
    cfg = oracle_cfg(VITL, 2)

    def dev_state():
        return {k: ({n: t.to(DEV) for n, t in v.items()} if k != "opt" else {}) for k, v in state.items()}
    res = {}
    for name, ctx in (("fp32", None), ("autocast-bf16", torch.autocast("cuda", dtype=torch.bfloat16))):
        st = dev_state()
        times = []
        for rep in range(2):   # second repetition is timed (first one pays allocator / kernel-selection warm-up)
            st_rep = {k: ({n: t.clone() for n, t in v.items()} if k != "opt" else {}) for k, v in st.items()}
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            if ctx is None:
                ref = O.train_step(st_rep, cd, med, mpd, cfg, dict(HP), 1)
            else:
                with ctx:
                    ref = O.train_step(st_rep, cd, med, mpd, cfg, dict(HP), 1)
            torch.cuda.synchronize()
            times.append(time.perf_counter() - t0)
        res[name] = (ref, times[-1])
        del st, st_rep
        torch.cuda.empty_cache()
    ref32 = res["fp32"][0]
    out = tr.train_step(cd, med, mpd, lr=ref32["lr"], wd=ref32["wd"], ema=ref32["ema"])
    print(f"ViT-L B=24 first step: HIP loss {out.loss:.6f} | eager fp32 {ref32['loss']:.6f} ({24 / res['fp32'][1]:.1f} clips/s)"
          f" | eager autocast-bf16 {res['autocast-bf16'][0]['loss']:.6f} ({24 / res['autocast-bf16'][1]:.1f} clips/s)")
    assert abs(out.loss - ref32["loss"]) < 1e-3 * abs(ref32["loss"]), (out.loss, ref32["loss"])
    assert abs(out.loss - res["autocast-bf16"][0]["loss"]) < 1e-3 * abs(ref32["loss"])
    # arena-wide: every gradient tensor of the step at the benched size, rel-L2 <= 3e-2 (measured 6e-3 .. 1.4e-2 in round 2)
    _arena_wide_gradient_check(tr, ref32["grads"], 3e-2, "ViT-L/16 B=24 vs GPU-eager fp32 oracle")


@pytest.mark.timeout(900)
def test_vit_huge_384_long_sequence_step_vs_gpu_eager_oracle():
    """BASELINE configs[4] shape (ViT-H/16, 16x384x384 -> 4608 tokens, head_dim 80: the long-sequence attention path and
    the 96-wide attention class inside a whole step), B=2, against the oracle run in fp32 by eager PyTorch on the same GPU:
    loss <= 1e-3 relative, EVERY gradient tensor rel-L2 <= 3e-2."""
    from oracle import vjepa_oracle as O
    from tests.step_util import VITH, VITL_MASKS
    m = dict(VITH, crop=384, num_patches=8 * 24 * 24)
    tr, state, _, _, _ = build_trainer(m, 2)
    gens = O.make_mask_gens(VITL_MASKS, m["crop"], m["frames"], m["patch"], m["tubelet"])
    clips, me, mp = draw_batch(gens, 2, m, 77, 78)
    cd, med, mpd = to_dev(clips, me, mp)
    st = {k: ({n: t.to(DEV) for n, t in v.items()} if k != "opt" else {}) for k, v in state.items()}
    ref = O.train_step(st, cd, med, mpd, oracle_cfg(m, 2), dict(HP), 1)
    out = tr.train_step(cd, med, mpd, lr=ref["lr"], wd=ref["wd"], ema=ref["ema"])
    assert me[0].shape[1] + mp[0].shape[1] > 2500, "test setup: a long predictor sequence"
    assert abs(out.loss - ref["loss"]) < 1e-3 * abs(ref["loss"]), (out.loss, ref["loss"])
    worst, worst_name = _arena_wide_gradient_check(tr, ref["grads"], 3e-2, "ViT-H/16 16x384x384 B=2 vs GPU-eager fp32 oracle")
    print(f"ViT-H 16x384x384 B=2: HIP loss {out.loss:.6f} vs GPU-eager fp32 oracle {ref['loss']:.6f}; worst gradient rel-L2 {worst:.2e}; "
          f"sequence lengths enc {[x.shape[1] for x in me]} pred {[x.shape[1] for x in mp]}")


@pytest.mark.timeout(900)
def test_vit_large_ten_step_loss_curve_vs_gpu_eager_oracle():
    """Ten consecutive optimisation steps of the BASELINE model (ViT-L/16, 16x224x224, vitl16.yaml masks and schedule
    shape, B=4): the bf16 HIP trajectory (AdamW, EMA, schedules included) stays within 1e-3 relative of the fp32 trajectory
    of the oracle, executed by eager PyTorch on the same GPU, at EVERY step; trained weights end within 5e-3 rel-L2 of the
    oracle's (the first Adam steps move every weight by ~lr * sign(g): a flipped sign on a near-zero gradient costs 2*lr,
    |w| ~ 0.02, ten steps at lr 2e-4 .. 6e-4) and the EMA target within 2e-4."""
    from oracle import vjepa_oracle as O
    from tests.step_util import VITL, VITL_MASKS
    tr, state, _, _, _ = build_trainer(VITL, 2)
    gens = O.make_mask_gens(VITL_MASKS, VITL["crop"], VITL["frames"], VITL["patch"], VITL["tubelet"])
    st = {k: ({n: t.to(DEV) for n, t in v.items()} if k != "opt" else {}) for k, v in state.items()}
    cfg = oracle_cfg(VITL, 2)
    hp = dict(HP, ipe=20, warmup=0.25)   # 5 warm-up steps, then the cosine part: both schedule branches are exercised
    worst = 0.0
    for step in range(1, 11):
        clips, me, mp = draw_batch(gens, 4, VITL, 500 + step, 900 + step)
        cd, med, mpd = to_dev(clips, me, mp)
        ref = O.train_step(st, cd, med, mpd, cfg, hp, step)
        out = tr.train_step(cd, med, mpd, lr=ref["lr"], wd=ref["wd"], ema=ref["ema"])
        rel = abs(out.loss - ref["loss"]) / abs(ref["loss"])
        worst = max(worst, rel)
        assert rel < 1e-3, (step, out.loss, ref["loss"])
    for name in ("blocks.0.attn.qkv.weight", "blocks.23.mlp.fc2.weight"):
        w = tr.arena.f32("enc." + name)
        assert float((w - st["enc"][name]).norm() / st["enc"][name].norm()) < 5e-3, name
        t = tr.tarena.f32("enc." + name)
        assert float((t - st["tgt"][name]).norm() / st["tgt"][name].norm()) < 2e-4, name
    print(f"ViT-L B=4, 10 steps: worst per-step relative loss deviation {worst:.2e}")

