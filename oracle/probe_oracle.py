"""ORACLE -- test infrastructure only.  CPU fp32 restatement of the attentive probe that the reference trains on frozen
V-JEPA features (evals/video_classification_frozen/eval.py:200-212 builds `AttentiveClassifier(embed_dim, num_heads, depth=1,
num_classes)`; its forward/backward is what row f4 of SURVEY.md section 8 widens into).

Restated as plain functions over a flat {name: tensor} weight dict with the reference's state-dict names.  Only tests/ may
import this module; nothing under jepa_amd/ does.  Pinned by tests/test_probe_oracle.py against tests/golden/probe.npz, which
oracle/make_golden_probe.py generates from the REAL reference module (src/models/attentive_pooler.py) in the build container.
"""
import torch
import torch.nn.functional as F


def cross_attention(w, pre, q, x, heads):
    """CrossAttention.forward, src/models/utils/modules.py:140-157.  q [B,n,C], x [B,N,C].  The module owns a `proj`
    Linear that its forward never applies (modules.py:156-157 return right after the head merge): not applied here either."""
    B, n, C = q.shape
    N = x.shape[1]
    hd = C // heads
    qh = F.linear(q, w[pre + "q.weight"], w.get(pre + "q.bias")).reshape(B, n, heads, hd).permute(0, 2, 1, 3)
    kv = F.linear(x, w[pre + "kv.weight"], w.get(pre + "kv.bias")).reshape(B, N, 2, heads, hd).permute(2, 0, 3, 1, 4)
    k, v = kv[0], kv[1]
    att = torch.softmax((qh @ k.transpose(-2, -1)) * hd ** -0.5, dim=-1)   # SDPA default scale = head_dim^-0.5
    return (att @ v).transpose(1, 2).reshape(B, n, C)


def cross_attention_block(w, pre, q, x, heads, eps):
    """CrossAttentionBlock.forward, modules.py:177-181: q + xattn(q, norm1(x)), then + mlp(norm2(q))."""
    C = x.shape[-1]
    y = cross_attention(w, pre + "xattn.", q, F.layer_norm(x, (C,), w[pre + "norm1.weight"], w[pre + "norm1.bias"], eps), heads)
    q = q + y
    h = F.layer_norm(q, (C,), w[pre + "norm2.weight"], w[pre + "norm2.bias"], eps)
    h = F.linear(F.gelu(F.linear(h, w[pre + "mlp.fc1.weight"], w[pre + "mlp.fc1.bias"])), w[pre + "mlp.fc2.weight"],
                 w[pre + "mlp.fc2.bias"])                                      # MLP.forward, modules.py:30-36 (exact-erf GELU)
    return q + h


def attentive_classifier(w, x, heads, eps=1e-5):
    """AttentiveClassifier.forward (attentive_pooler.py:132-135) over AttentivePooler.forward (96-102) with depth = 1,
    complete_block = True, one query token; nn.LayerNorm default eps = 1e-5 (the classifier is built with the default
    norm_layer, eval.py:205-210)."""
    q = w["pooler.query_tokens"].repeat(len(x), 1, 1)
    q = cross_attention_block(w, "pooler.cross_attention_block.", q, x, heads, eps)
    return F.linear(q.squeeze(1), w["linear.weight"], w["linear.bias"])


def probe_loss_and_grads(w, x, labels, heads, eps=1e-5):
    """Cross-entropy of the probe (eval.py:298-300 `criterion = torch.nn.CrossEntropyLoss()`), logits and the gradient of every
    parameter the forward uses (the never-applied xattn.proj.* keep grad None in the reference and are absent here)."""
    ws = {k: v.detach().clone().requires_grad_(True) for k, v in w.items()}
    logits = attentive_classifier(ws, x, heads, eps)
    loss = F.cross_entropy(logits, labels)
    loss.backward()
    return loss.detach(), logits.detach(), {k: v.grad for k, v in ws.items() if v.grad is not None}
