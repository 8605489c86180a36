"""Pin the oracle (oracle/vjepa_oracle.py, a CPU restatement) against the fixtures produced by the REAL
reference modules (tests/golden/*.npz, oracle/make_golden.py).  CPU only."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import vjepa_oracle as O  # noqa: E402
from tests.golden_util import HP, MICRO, MICRO_MASKS, load_micro, load_tables, micro_weights, rel_l2, step_inputs  # noqa: E402


def test_sincos_tables_bit_exact_in_fp32():
    t = load_tables()
    for D, gs, gd in [(192, 4, 4), (96, 4, 4), (64, 4, 4), (32, 4, 4)]:
        ours = O.sincos_3d(D, gs, gd, uniform_power=True)
        ref = t[f"pos3d_{D}_{gs}_{gd}"]
        assert ours.shape == ref.shape
        assert np.array_equal(ours.astype(np.float32), ref.astype(np.float32))
        assert np.abs(ours - ref).max() < 1e-15
    for D, gs, gd, up in [(1024, 14, 8, True), (384, 14, 8, True), (1280, 24, 8, True), (768, 14, 8, False)]:
        ours = O.sincos_3d(D, gs, gd, uniform_power=up)
        assert np.array_equal(ours[::53].astype(np.float32), t[f"pos3d_rows53_{D}_{gs}_{gd}_{int(up)}"].astype(np.float32))
        sums = np.array([ours.sum(), (ours * ours).sum(), np.abs(ours).sum()])
        assert np.allclose(sums, t[f"pos3d_sums_{D}_{gs}_{gd}_{int(up)}"], rtol=1e-13)


def test_collator_indices_bit_exact():
    t = load_tables()
    vitl = [dict(aspect_ratio=(0.75, 1.5), num_blocks=8, spatial_scale=(0.15, 0.15), temporal_scale=(1.0, 1.0)),
            dict(aspect_ratio=(0.75, 1.5), num_blocks=2, spatial_scale=(0.7, 0.7), temporal_scale=(1.0, 1.0))]
    gens = O.make_mask_gens(vitl, 224, 16, 16, 2)
    for it in range(3):
        torch.manual_seed(4321 + it)
        for i, g in enumerate(gens):
            e, p = g(6)
            assert e.dtype == torch.int64
            assert np.array_equal(e.numpy(), t[f"vitl_it{it}_enc{i}"])
            assert np.array_equal(p.numpy(), t[f"vitl_it{it}_pred{i}"])
    odd = [dict(aspect_ratio=(0.3, 3.0), num_blocks=3, spatial_scale=(0.2, 0.8), temporal_scale=(0.25, 1.0),
                max_temporal_keep=0.5, max_keep=100)]
    g = O.make_mask_gens(odd, 224, 16, 16, 2)[0]
    for it in range(3):
        torch.manual_seed(99 + it)
        e, p = g(4)
        assert np.array_equal(e.numpy(), t[f"odd_it{it}_enc0"])
        assert np.array_equal(p.numpy(), t[f"odd_it{it}_pred0"])


def test_schedules():
    t = load_tables()
    T = int(1.25 * 300 * 300)
    for s, lr, wd in zip(t["sched_steps"], t["sched_lr"], t["sched_wd"]):
        assert abs(O.lr_at(int(s), 40 * 300, 2e-4, 6.25e-4, 1e-6, T) - lr) <= 1e-18 + 1e-12 * lr
        assert abs(O.wd_at(int(s), 0.04, 0.4, T) - wd) <= 1e-12 * wd


def test_micro_two_steps_match_reference():
    z = load_micro()
    enc, pred = micro_weights(z)
    state = dict(enc=enc, pred=pred, tgt={k: v.clone() for k, v in enc.items()}, opt={})
    gens_seen = []
    for s in range(2):
        clips, me, mp = step_inputs(z, s)
        out = O.train_step(state, clips, me, mp, MICRO, HP, s + 1)
        sc = z[f"s{s}/scalars"]
        assert abs(out["loss"] - sc[0]) < 2e-6 * abs(sc[0])
        assert abs(out["loss_jepa"] - sc[1]) < 2e-6 * abs(sc[1])
        assert abs(out["loss_reg"] - sc[2]) < 1e-5 * abs(sc[2]) + 1e-7
        assert abs(out["lr"] - sc[3]) < 1e-15 and abs(out["wd"] - sc[4]) < 1e-12 and abs(out["ema"] - sc[5]) < 1e-15
        for i in range(2):
            assert rel_l2(out["h"][i], z[f"s{s}/h{i}"]) < 1e-5
            assert rel_l2(out["z_enc"][i], z[f"s{s}/z_enc{i}"]) < 1e-5
            assert rel_l2(out["z"][i], z[f"s{s}/z{i}"]) < 1e-5
        for k in z.files:
            if k.startswith(f"s{s}/grad/"):
                _, _, grp, name = k.split("/", 3)
                assert rel_l2(out["grads"][grp][name], z[k]) < 2e-4, (k, rel_l2(out["grads"][grp][name], z[k]))
            if k.startswith(f"s{s}/post/"):
                _, _, grp, name = k.split("/", 3)
                assert rel_l2(state[grp][name], z[k]) < 1e-5, (k, rel_l2(state[grp][name], z[k]))
        gens_seen.append([m.shape[1] for m in me])
    # the collator restatement reproduces the fixture's mask draws too
    gens = O.make_mask_gens(MICRO_MASKS, MICRO["crop"], MICRO["frames"], MICRO["patch"], MICRO["tubelet"])
    for s in range(2):
        torch.manual_seed(4321 + s)
        for i, g in enumerate(gens):
            e, p = g(2)
            assert np.array_equal(e.numpy(), z[f"s{s}/masks_enc{i}"])
            assert np.array_equal(p.numpy(), z[f"s{s}/masks_pred{i}"])


def test_flop_model_matches_survey():
    cfg = dict(embed_dim=1024, depth=24, heads=16, pred_dim=384, pred_depth=12, patch=16, tubelet=2, num_patches=1568)
    f = O.step_flops(cfg, 1, [366, 107], [747, 1101])
    assert 2.4e12 < f < 2.8e12  # ~2.57 TFLOP per clip at the mean mask sizes
