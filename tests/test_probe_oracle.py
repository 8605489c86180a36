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
