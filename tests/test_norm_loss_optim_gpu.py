"""LayerNorm forward / backward (+ column sums of dx), the target rows + latent loss kernels (reference app/vjepa/train.py:424-446) and the fused
AdamW + EMA + bf16 re-cast (train.py:461-487) against fp32 PyTorch references."""
import math
import pytest
import torch
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.golden_util import rel_l2  # noqa: E402
from tests.step_util import TINY, TINY_MASKS, build_trainer, draw_batch, to_dev  # noqa: E402

pytestmark = pytest.mark.gpu


DEV = "cuda"


@pytest.fixture(scope="module")
def ops():
    from jepa_amd.hip import ops as _ops
    return _ops


def rel_l2(a, b):
    a, b = a.float(), b.float()
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


def bf(x):
    return x.to(torch.bfloat16)


# ------------------------------------------------------------------------------------------ layernorm
@pytest.mark.parametrize("rows,D", [(40, 192), (473, 1024), (1000, 384), (7, 1280), (5, 96)])
def test_layernorm_fwd_bwd(ops, rows, D):
    g = torch.Generator().manual_seed(5)
    x = bf(torch.randn(rows, D, generator=g) * 2 + 0.3).to(DEV)
    gamma = (1 + 0.1 * torch.randn(D, generator=g)).to(DEV)
    beta = (0.1 * torch.randn(D, generator=g)).to(DEV)
    dy = bf(torch.randn(rows, D, generator=g)).to(DEV)
    dres = bf(torch.randn(rows, D, generator=g)).to(DEV)
    eps = 1e-6
    y, mean, rstd = ops.layernorm_fwd(x, gamma, beta, eps)
    xr = x.float().requires_grad_(True)
    gr = gamma.clone().requires_grad_(True)
    br = beta.clone().requires_grad_(True)
    yr = torch.nn.functional.layer_norm(xr, (D,), gr, br, eps)
    assert rel_l2(y, yr) < 4e-3, rel_l2(y, yr)
    assert torch.allclose(mean, x.float().mean(-1), rtol=1e-4, atol=1e-5)
    yr.backward(dy.float())
    dgamma = torch.zeros(D, device=DEV)
    dbeta = torch.zeros(D, device=DEV)
    dx = ops.layernorm_bwd(dy, x, gamma, mean, rstd, dgamma, dbeta, dres=dres)
    assert rel_l2(dx, xr.grad + dres.float()) < 6e-3, rel_l2(dx, xr.grad + dres.float())
    assert rel_l2(dgamma, gr.grad) < 1e-4, rel_l2(dgamma, gr.grad)
    assert rel_l2(dbeta, br.grad) < 1e-4, rel_l2(dbeta, br.grad)
    # accumulate + alpha
    dx2 = ops.layernorm_bwd(dy, x, gamma, mean, rstd, dgamma, dbeta, alpha=2.0, accumulate=True)
    assert rel_l2(dgamma, 3 * gr.grad) < 1e-4
    assert rel_l2(dx2, xr.grad) < 6e-3


# ------------------------------------------------------------------------------------------------ LayerNorm backward + column sums
@pytest.mark.parametrize("rows,D", [(10560, 1024), (5533, 384), (777, 1280), (40, 192)])
def test_layernorm_bwd_column_sums_of_dx(rows, D):
    """vj_layernorm_bwd_colsum: dx / dgamma / dbeta equal to the plain backward (up to fp contraction in a separately
    compiled variant); dxsum = alpha * column sums of dx
    (accumulated in fp32 BEFORE the bf16 rounding of dx) within 2e-3 relative of the fp32 sum of the rounded dx, and
    accumulating (beta = 1) adds to the previous value."""
    from jepa_amd.hip import ops
    g = torch.Generator(device=DEV).manual_seed(rows + D)
    x = torch.randn(rows, D, device=DEV, generator=g).to(torch.bfloat16)
    dy = torch.randn(rows, D, device=DEV, generator=g).to(torch.bfloat16)
    dres = torch.randn(rows, D, device=DEV, generator=g).to(torch.bfloat16)
    gamma = torch.randn(D, device=DEV, generator=g)
    beta = torch.randn(D, device=DEV, generator=g)
    _, mean, rstd = ops.layernorm_fwd(x, gamma, beta, 1e-6)
    outs = []
    for with_cs in (False, True):
        dg, db = torch.zeros(D, device=DEV), torch.zeros(D, device=DEV)
        cs = torch.full((D,), 3.0, device=DEV) if with_cs else None
        dx = ops.layernorm_bwd(dy, x, gamma, mean, rstd, dg, db, dres=dres, alpha=0.5, dxsum=cs)
        if with_cs:
            first = cs.clone()
            ops.layernorm_bwd(dy, x, gamma, mean, rstd, dg, db, dres=dres, alpha=0.5, accumulate=True, dxsum=cs)
            assert torch.allclose(cs, 2 * first, rtol=1e-6, atol=1e-6)
            cs = first
        outs.append((dx, dg.clone(), db.clone(), cs))
    # the two template variants are compiled separately under -ffp-contract=fast: dx may differ by one bf16 ulp in a few
    # elements, never more (C chain and Python chain call the same variant for the same tensor: their bit-identity holds)
    da, dbb = outs[0][0].float(), outs[1][0].float()
    frac = float((da != dbb).float().mean())
    print(f"layernorm_bwd colsum variant vs plain, rows={rows} D={D}: {100 * frac:.4f} % of dx elements differ, rel-L2 "
          f"{float((da - dbb).norm() / da.norm()):.2e}")
    assert frac < 2e-3 and float((da - dbb).norm() / da.norm()) < 2e-4
    ref = 0.5 * outs[1][0].float().sum(0)
    err = float((outs[1][3] - ref).norm() / ref.norm())
    assert err < 2e-3, err
    # the first call of the with_cs arm ran with beta = 0: dgamma/dbeta equal the plain call's halves after the accumulate
    assert torch.allclose(outs[1][1], 2 * outs[0][1], rtol=1e-6, atol=1e-5)
    assert torch.allclose(outs[1][2], 2 * outs[0][2], rtol=1e-6, atol=1e-5)


def test_target_rows_and_loss(ops):
    g = torch.Generator().manual_seed(12)
    B, N, K, D = 2, 64, 32, 192
    x = bf(torch.randn(B * N, D, generator=g) * 3).to(DEV)
    gamma = (1 + 0.1 * torch.randn(D, generator=g)).to(DEV)
    beta = (0.1 * torch.randn(D, generator=g)).to(DEV)
    idx = torch.stack([torch.randperm(N, generator=g)[:K].sort().values for _ in range(B)]).to(DEV)
    h = ops.target_rows(x, gamma, beta, idx, B, N, 1e-6)
    F = torch.nn.functional
    hr = F.layer_norm(F.layer_norm(x.float().view(B, N, D), (D,), gamma, beta, 1e-6), (D,))
    hr = torch.gather(hr, 1, idx.unsqueeze(-1).repeat(1, 1, D))
    assert torch.allclose(h, hr, rtol=1e-4, atol=1e-4), (h - hr).abs().max()
    z = bf(torch.randn(B * K, D, generator=g)).to(DEV)
    loss = torch.zeros(1, device=DEV)
    dz = torch.empty_like(z)
    numel = z.numel()
    ops.latent_loss(z, h, loss, p=1.0, out_scale=1.0 / numel, dz=dz, gscale=0.25)
    ref = (z.float().view(B, K, D) - hr).abs().mean()
    assert abs(loss.item() - ref.item()) < 1e-5 * abs(ref.item()) + 1e-7
    assert torch.equal(dz.float().view(B, K, D), 0.25 * torch.sign(z.float().view(B, K, D) - h))
    loss2 = torch.ones(1, device=DEV)
    ops.latent_loss(z, h, loss2, p=2.0, out_scale=1.0 / numel, accumulate=True)
    ref2 = 1.0 + ((z.float().view(B, K, D) - hr).abs() ** 2).mean() / 2
    assert abs(loss2.item() - ref2.item()) < 1e-4 * ref2.item()
    # variance regulariser (reg_fn)
    pstd = torch.zeros(B, D, device=DEV)
    ops.token_pstd(z, pstd, B, K, D, False)
    ref_p = torch.sqrt(z.float().view(B, K, D).var(dim=1) + 1e-4)
    assert torch.allclose(pstd, ref_p, rtol=1e-4, atol=1e-5)
    out = torch.zeros(1, device=DEV)
    ops.reg_finish(pstd, 1, out)
    assert abs(out.item() - torch.relu(1 - ref_p).mean().item()) < 1e-5


def test_adamw_ema_matches_torch(ops):
    g = torch.Generator().manual_seed(13)
    n = 4096 + 64
    p0 = torch.randn(n, generator=g)
    tgt0 = p0.clone()
    pr = torch.nn.Parameter(p0.clone())
    opt = torch.optim.AdamW([pr], lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.05)
    p = p0.clone().to(DEV)
    m = torch.zeros(n, device=DEV)
    v = torch.zeros(n, device=DEV)
    tgt = tgt0.clone().to(DEV)
    pb = torch.empty(n, dtype=torch.bfloat16, device=DEV)
    tb = torch.empty(n, dtype=torch.bfloat16, device=DEV)
    tr = tgt0.clone()
    for step in range(1, 4):
        grad = torch.randn(n, generator=g)
        pr.grad = grad.clone()
        opt.step()
        tr.mul_(0.998).add_((1 - 0.998) * pr.detach())
        ops.adamw_ema(p, grad.to(DEV), m, v, pb, tgt, tb, 1e-3, 0.05, 0.9, 0.999, 1e-8, step, 1.0, 0.998)
    assert torch.allclose(p.cpu(), pr.detach(), rtol=1e-5, atol=1e-6), (p.cpu() - pr.detach()).abs().max()
    assert torch.allclose(tgt.cpu(), tr, rtol=1e-5, atol=1e-6)
    assert torch.equal(pb, bf(p))
    assert torch.equal(tb, bf(tgt))
    out2 = torch.zeros(2, device=DEV)
    ops.sqnorm(p, out2)
    assert abs(out2[0].item() - (p.double() ** 2).sum().item()) < 1e-4 * out2[0].item()
    assert out2[1].item() == 0
    p[3] = float("nan")
    ops.sqnorm(p, out2)
    assert out2[1].item() == 1

