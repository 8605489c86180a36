"""Checkpoint format round trip INTO the reference, on CPU.

A checkpoint dictionary written the way jepa_amd.app.vjepa.train.save_checkpoint writes it (state-dict keys
`module.backbone.*`, optimizer state from engine/optstate.py) is loaded by the reference's OWN code:
  * app/vjepa/utils.py:28-83   load_checkpoint  -> reference modules + a torch.optim.AdamW built by the reference's init_opt
  * evals/video_classification_frozen/eval.py:414-441  load_pretrained (the `module.` / `backbone.` key stripper)
and a CPU forward of the loaded reference encoder is compared with the pinned oracle on the same weights.
The reference-dependent tests carry the `reference` marker (skipped where /root/reference is absent, e.g. the GPU box);
the id / group-size contract of the optimizer state is also checked without the reference against a torch AdamW built
from the documented grouping."""
import ast
import copy
import logging
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.step_util import TINY, build_models, oracle_cfg  # noqa: E402

REF = "/root/reference"


def _fake_trained_state(enc, pred, step=3, seed=0):
    """Arena layout + random Adam moments on CPU, exactly as the Trainer lays them out (engine/optstate.py)."""
    from jepa_amd.engine import optstate as OS
    enc_named = list(enc.named_parameters())
    pred_named = list(pred.named_parameters())
    train = lambda named, nodecay: [(("e." if named is enc_named else "p.") + n, p) for n, p in named   # noqa: E731
                                    if p.requires_grad and OS.is_no_decay(n, p) == nodecay]
    slots, ranges, total = OS.layout([train(enc_named, False), train(enc_named, True), train(pred_named, False),
                                      train(pred_named, True)])
    g = torch.Generator().manual_seed(seed)
    M1, M2 = torch.randn(total, generator=g), torch.rand(total, generator=g)
    slot_of = {id(s.param): s for s in slots.values()}
    groups = [dict(g_, params=[p for _, p in ps], lr=1e-3, weight_decay=g_.get("weight_decay", 0.04))
              for g_, ps in OS.reference_groups(enc_named, pred_named)]
    sd = OS.build_state_dict(groups, slot_of, M1, M2, step, (0.9, 0.999), 1e-8)
    return sd, groups, slot_of, M1, M2


def _checkpoint(enc, pred, tgt, opt_sd, epoch=7):
    pre = lambda sd: {"module." + k: v.clone() for k, v in sd.items()}   # noqa: E731  (train.py: _with_module_prefix)
    return {"encoder": pre(enc.state_dict()), "predictor": pre(pred.state_dict()), "opt": opt_sd, "scaler": None,
            "target_encoder": pre(tgt.state_dict()), "epoch": epoch, "loss": 0.5, "batch_size": 2, "world_size": 1,
            "lr": 1e-3}


def test_optimizer_state_ids_match_a_torch_adamw_built_like_init_opt():
    """No reference needed: groups = [enc >=2-D non-bias, pred >=2-D non-bias, enc bias/1-D, pred bias/1-D] over ALL
    named parameters (frozen sincos tables included) -- torch.optim.AdamW.load_state_dict validates ids and sizes."""
    from jepa_amd.engine import optstate as OS
    enc, pred = build_models(TINY, 2)
    sd, groups, slot_of, M1, M2 = _fake_trained_state(enc, pred)
    opt = torch.optim.AdamW([dict(g, params=list(g["params"])) for g in groups])
    opt.load_state_dict(sd)
    n = 0
    for g in opt.param_groups:
        for p in g["params"]:
            if p.requires_grad:
                s = slot_of[id(p)]
                assert torch.equal(opt.state[p]["exp_avg"].reshape(-1), M1[s.off:s.off + s.numel])
                assert torch.equal(opt.state[p]["exp_avg_sq"].reshape(-1), M2[s.off:s.off + s.numel])
                assert float(opt.state[p]["step"]) == 3.0
                n += 1
            else:
                assert p not in opt.state
    assert n == len(slot_of)
    # and back: load what torch wrote into zeroed arenas
    Z1, Z2 = torch.zeros_like(M1), torch.zeros_like(M2)
    step = OS.load_state_dict(groups, slot_of, Z1, Z2, opt.state_dict())
    used = torch.zeros_like(M1, dtype=torch.bool)
    for s in slot_of.values():
        used[s.off:s.off + s.numel] = True
    assert step == 3 and torch.equal(Z1[used], M1[used]) and torch.equal(Z2[used], M2[used])


def _ref_imports():
    if REF not in sys.path:
        sys.path.insert(0, REF)
    import app.vjepa.utils as RU            # noqa: E402  (reference module, read-only)
    return RU


class _DDPLike(torch.nn.Module):
    """The reference checkpoints DistributedDataParallel-wrapped modules: their state-dict keys start with `module.`."""

    def __init__(self, m):
        super().__init__()
        self.module = m


@pytest.mark.reference
def test_reference_load_checkpoint_reads_a_jepa_amd_checkpoint(tmp_path):
    RU = _ref_imports()
    enc, pred = build_models(TINY, 2, perturb_small=True)
    tgt = copy.deepcopy(enc)
    with torch.no_grad():
        for p in tgt.parameters():
            p.mul_(0.97)            # the EMA target differs from the encoder in a real checkpoint
    opt_sd, groups, slot_of, M1, M2 = _fake_trained_state(enc, pred, step=5)
    path = os.path.join(tmp_path, "jepa-latest.pth.tar")
    torch.save(_checkpoint(enc, pred, tgt, opt_sd, epoch=7), path)
    # fresh reference modules (different seed -> different weights) + the reference's own optimizer
    torch.manual_seed(123)
    r_enc, r_pred = RU.init_video_model(device="cpu", patch_size=16, num_frames=TINY["frames"], tubelet_size=2,
                                        model_name="vit_tiny", crop_size=TINY["crop"], pred_depth=TINY["pred_depth"],
                                        pred_embed_dim=TINY["pred_dim"], uniform_power=True, use_mask_tokens=True,
                                        num_mask_tokens=2, zero_init_mask_tokens=True, use_sdpa=True)
    r_tgt = copy.deepcopy(r_enc)
    r_opt, r_scaler, _, _ = RU.init_opt(encoder=r_enc, predictor=r_pred, wd=0.04, final_wd=0.4, start_lr=2e-4, ref_lr=6e-4,
                                        final_lr=1e-6, iterations_per_epoch=10, warmup=1, num_epochs=4, ipe_scale=1.25,
                                        mixed_precision=False, betas=(0.9, 0.999), eps=1e-8)
    w_enc, w_pred, w_tgt = _DDPLike(r_enc), _DDPLike(r_pred), _DDPLike(r_tgt)
    *_, epoch = RU.load_checkpoint(r_path=path, encoder=w_enc, predictor=w_pred, target_encoder=w_tgt, opt=r_opt,
                                   scaler=r_scaler)
    assert epoch == 7, "the reference's load_checkpoint swallowed an exception and restarted at epoch 0"
    for ours, theirs in ((enc, r_enc), (pred, r_pred), (tgt, r_tgt)):
        a, b = ours.state_dict(), theirs.state_dict()
        assert list(a.keys()) == list(b.keys())
        for k in a:
            assert torch.equal(a[k], b[k]), k
    # optimizer moments landed on the right reference parameters (name-by-name through the shared key order)
    ours_by_name = {("e." if m is enc else "p.") + n: p for m in (enc, pred) for n, p in m.named_parameters()}
    for m, r in ((enc, r_enc), (pred, r_pred)):
        for (n, p), (rn, rp) in zip(m.named_parameters(), r.named_parameters()):
            assert n == rn
            if p.requires_grad:
                s = slot_of[id(p)]
                assert torch.equal(r_opt.state[rp]["exp_avg"].reshape(-1), M1[s.off:s.off + s.numel]), n
                assert float(r_opt.state[rp]["step"]) == 5.0
            else:
                assert rp not in r_opt.state
    assert len(ours_by_name) == len(list(r_enc.named_parameters())) + len(list(r_pred.named_parameters()))


def _reference_function(rel_path, name, extra_globals):
    """Execute ONE function definition out of a reference file that cannot be imported here (its module pulls in
    torchvision / decord): the reference's own source, read in place, never copied."""
    src = open(os.path.join(REF, rel_path)).read()
    node = next(n for n in ast.parse(src).body if isinstance(n, ast.FunctionDef) and n.name == name)
    ns = dict(extra_globals)
    exec(compile(ast.Module(body=[node], type_ignores=[]), rel_path, "exec"), ns)
    return ns[name]


@pytest.mark.reference
def test_reference_eval_loader_and_cpu_forward(tmp_path, capsys):
    """evals/video_classification_frozen/eval.py:414-441 strips `module.` / `backbone.` and loads the TARGET encoder into
    a bare reference VisionTransformer; its CPU forward equals the pinned oracle's on the same weights (1e-5)."""
    _ref_imports()
    import src.models.vision_transformer as RV            # noqa: E402  (reference)
    from oracle import vjepa_oracle as O
    enc, pred = build_models(TINY, 2, perturb_small=True)
    tgt = copy.deepcopy(enc)
    with torch.no_grad():
        for p in tgt.parameters():
            p.mul_(0.97)
    opt_sd, *_ = _fake_trained_state(enc, pred)
    path = os.path.join(tmp_path, "jepa-latest.pth.tar")
    torch.save(_checkpoint(enc, pred, tgt, opt_sd, epoch=3), path)
    load_pretrained = _reference_function("evals/video_classification_frozen/eval.py", "load_pretrained",
                                          {"torch": torch, "logger": logging.getLogger("ref-eval")})
    torch.manual_seed(99)
    bare = RV.vit_tiny(img_size=TINY["crop"], patch_size=16, num_frames=TINY["frames"], tubelet_size=2,
                       uniform_power=True, use_sdpa=True)
    bare = load_pretrained(encoder=bare, pretrained=path, checkpoint_key="target_encoder")
    capsys.readouterr()   # the reference prints the whole model
    for k, v in tgt.backbone.state_dict().items():
        assert torch.equal(bare.state_dict()[k], v), k
    clips = torch.randn(2, 3, TINY["frames"], TINY["crop"], TINY["crop"], generator=torch.Generator().manual_seed(1))
    with torch.no_grad():
        y_ref = bare(clips)
    w = {k: v.detach().clone() for k, v in tgt.backbone.state_dict().items()}
    y_or = O.encoder_forward(w, clips, oracle_cfg(TINY, 2))
    assert (y_ref - y_or).abs().max() <= 1e-5 * y_or.abs().max() + 1e-6
