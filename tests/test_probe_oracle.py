"""CPU: the attentive-probe oracle (oracle/probe_oracle.py) against the fixture generated from the real reference
AttentiveClassifier (tests/golden/probe.npz <- oracle/make_golden_probe.py).  fp32 on both sides: tight tolerances."""
import os

import numpy as np
import pytest
import torch

from oracle import probe_oracle as po

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "probe.npz")


def load_case(name):
    d = np.load(GOLD)
    cfg = [int(v) for v in d[f"{name}.cfg"]]
    w = {k[len(name) + 3:]: torch.from_numpy(d[k]) for k in d.files if k.startswith(f"{name}.w.")}
    g = {k[len(name) + 3:]: torch.from_numpy(d[k]) for k in d.files if k.startswith(f"{name}.g.")}
    return cfg, w, g, torch.from_numpy(d[f"{name}.x"]), torch.from_numpy(d[f"{name}.labels"]), \
        torch.from_numpy(d[f"{name}.logits"]), float(d[f"{name}.loss"])


@pytest.mark.parametrize("name", ["a", "b"])
def test_probe_oracle_matches_reference_fixture(name):
    (B, N, D, H, C), w, g, x, labels, logits, loss = load_case(name)
    o_loss, o_logits, o_grads = po.probe_loss_and_grads(w, x, labels, H)
    assert torch.allclose(o_logits, logits, rtol=1e-5, atol=1e-6), (o_logits - logits).abs().max()
    assert abs(float(o_loss) - loss) < 1e-6 * max(1.0, abs(loss))
    # the reference leaves the never-applied xattn.proj without gradient; every other parameter must agree
    assert set(o_grads) == set(g), set(o_grads) ^ set(g)
    assert "pooler.cross_attention_block.xattn.proj.weight" in w and "pooler.cross_attention_block.xattn.proj.weight" not in g
    for k in g:
        assert torch.allclose(o_grads[k], g[k], rtol=1e-4, atol=1e-7), (k, (o_grads[k] - g[k]).abs().max())


def test_probe_module_keeps_the_reference_names_and_has_no_cpu_path():
    """Drop-in check without a GPU: the HIP-backed AttentiveClassifier exposes exactly the reference's parameter names and shapes
    (the fixture's state dict loads strictly), re-initialises like the reference (LayerNorm (1, 0), zero biases, proj / fc2 rescaled
    by 1/sqrt(2)), refuses unsupported constructor settings and raises on CPU tensors instead of falling back."""
    from jepa_amd.src.models.attentive_pooler import AttentiveClassifier, AttentivePooler
    (B, N, D, H, C), w, g, x, labels, logits, loss = load_case("a")
    m = AttentiveClassifier(embed_dim=D, num_heads=H, depth=1, num_classes=C)
    assert {n: tuple(p.shape) for n, p in m.named_parameters()} == {k: tuple(v.shape) for k, v in w.items()}
    blk = m.pooler.cross_attention_block
    assert float(blk.norm1.weight.min()) == 1.0 and float(blk.norm1.bias.abs().max()) == 0.0
    assert float(blk.xattn.kv.bias.abs().max()) == 0.0   # (the head `linear` keeps nn.Linear's default init, as in the reference)
    # trunc_normal(std=0.02), then / sqrt(2) for fc2 and the (never applied) proj (attentive_pooler.py:68-81)
    assert abs(float(blk.mlp.fc1.weight.std()) - 0.02) < 0.002 and abs(float(blk.xattn.kv.weight.std()) - 0.02) < 0.002
    assert abs(float(blk.mlp.fc2.weight.std()) - 0.02 / 2 ** 0.5) < 0.0015
    assert abs(float(blk.xattn.proj.weight.std()) - 0.02 / 2 ** 0.5) < 0.0015
    m.load_state_dict(w, strict=True)
    with pytest.raises(ValueError):
        m(x)
    for kw in (dict(num_queries=2), dict(depth=2), dict(complete_block=False)):
        with pytest.raises(NotImplementedError):
            AttentivePooler(embed_dim=D, num_heads=H, **kw)
