"""Generate tests/golden/probe.npz from the REAL reference AttentiveClassifier (src/models/attentive_pooler.py:105-136), CPU fp32.

    python oracle/make_golden_probe.py            # needs /root/reference (build container only)

Contents: the state dict (seeded init), one batch of features / labels, the logits, the loss, and the gradient of every
parameter after `CrossEntropyLoss()(classifier(x), labels).backward()` -- what evals/video_classification_frozen/eval.py:330-352
computes for its probe.  The script is deterministic (re-running it reproduces the file bit for bit).
"""
import os
import sys

import numpy as np
import torch

REF = os.environ.get("JEPA_REFERENCE", "/root/reference")
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")
CASES = {   # name: (B, N, D, heads, classes)
    "a": (3, 50, 128, 2, 8),      # head_dim 64, N not a multiple of anything
    "b": (2, 129, 160, 2, 12),    # head_dim 80 (ViT-H class), N = 2 x 64 + 1
}


def main():
    sys.path.insert(0, REF)
    from src.models.attentive_pooler import AttentiveClassifier
    out = {}
    for name, (B, N, D, H, C) in CASES.items():
        torch.manual_seed(1234 + len(name) + N)
        m = AttentiveClassifier(embed_dim=D, num_heads=H, depth=1, num_classes=C)
        # the default init leaves every bias at 0 and every LayerNorm at (1, 0): perturb them so that their gradients and
        # their effect on the output are exercised
        g = torch.Generator().manual_seed(99 + N)
        with torch.no_grad():
            for n_, p in m.named_parameters():
                if p.dim() == 1:
                    p.add_(0.1 * torch.randn(p.shape, generator=g))
        x = torch.randn(B, N, D, generator=g)
        labels = torch.randint(0, C, (B,), generator=g)
        logits = m(x)
        loss = torch.nn.CrossEntropyLoss()(logits, labels)
        loss.backward()
        out[f"{name}.cfg"] = np.array([B, N, D, H, C], dtype=np.int64)
        out[f"{name}.x"] = x.numpy()
        out[f"{name}.labels"] = labels.numpy()
        out[f"{name}.logits"] = logits.detach().numpy()
        out[f"{name}.loss"] = loss.detach().numpy()
        for n_, p in m.named_parameters():
            out[f"{name}.w.{n_}"] = p.detach().numpy()
            if p.grad is not None:
                out[f"{name}.g.{n_}"] = p.grad.numpy()
    os.makedirs(OUT, exist_ok=True)
    np.savez_compressed(os.path.join(OUT, "probe.npz"), **out)
    print("wrote", os.path.join(OUT, "probe.npz"), {k: v.shape for k, v in out.items() if k.endswith(("logits", "loss"))})


if __name__ == "__main__":
    main()
