"""ORACLE -- test infrastructure only.  CPU fp32 restatement of the V-JEPA pretraining step.

This file restates, as plain functions over a flat {name: tensor} weight dict, the arithmetic the reference
(facebookresearch/jepa) executes for one `train_step` on a CPU-only box (autocast/GradScaler disable themselves
there, so everything is fp32).  It exists to CHECK the HIP path: only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may import it.  Nothing under jepa_amd/ imports this module, and it is never the
thing measured or shipped.

Pinning: the reference ships no tests or golden vectors ("parity unpinned" upstream).  This restatement is pinned
by tests/test_oracle_golden.py against tests/golden/micro_step.npz and tests/golden/host_tables.npz, which
oracle/make_golden.py generates from the *real* reference modules run in the build container (re-running that
script against /root/reference reproduces both files bit for bit).

Each function cites the reference lines it follows (paths relative to the reference root).
"""
import math

import numpy as np
import torch
import torch.nn.functional as F


# ------------------------------------------------------------------------------------------------------------
# positional embeddings -- src/models/utils/pos_embs.py:11-44, 81-99
# ------------------------------------------------------------------------------------------------------------
def sincos_1d(dim, pos):
    """pos_embs.py:81-99: [sin(pos*w) | cos(pos*w)], w_i = 10000^(-i/(dim/2)), float64."""
    assert dim % 2 == 0
    omega = 1.0 / 10000 ** (np.arange(dim // 2, dtype=np.float64) / (dim / 2.0))
    out = np.outer(np.asarray(pos, dtype=np.float64).reshape(-1), omega)
    return np.concatenate([np.sin(out), np.cos(out)], axis=1)


def sincos_3d(embed_dim, grid_size, grid_depth, uniform_power=False):
    """pos_embs.py:11-44: token order (d,h,w) row-major; channels [depth | height | width], truncated to D."""
    d, h, w = np.meshgrid(np.arange(grid_depth, dtype=np.float64), np.arange(grid_size, dtype=np.float64),
                          np.arange(grid_size, dtype=np.float64), indexing="ij")
    if uniform_power:
        hd = wd = dd = int(np.ceil(embed_dim / 6) * 2)  # pos_embs.py:35
    else:
        hd = wd = embed_dim // 4
        dd = embed_dim // 2
    emb = np.concatenate([sincos_1d(dd, d), sincos_1d(hd, h), sincos_1d(wd, w)], axis=1)
    return emb[:, :embed_dim]


# ------------------------------------------------------------------------------------------------------------
# multiblock mask collator -- src/masks/multiblock3d.py:66-203
# ------------------------------------------------------------------------------------------------------------
class MaskGenOracle:
    """One mask config.  RNG discipline follows the reference exactly: block size from a generator seeded with
    a step counter (multiblock3d.py:106-136,162-164), block positions from the global torch RNG (138-153)."""

    def __init__(self, crop_size, num_frames, patch, tubelet, spatial_scale, temporal_scale, aspect_ratio, npred,
                 max_temporal_keep=1.0, max_keep=None):
        self.h = self.w = crop_size // patch
        self.t = num_frames // tubelet
        self.spatial_scale, self.temporal_scale, self.aspect_ratio = spatial_scale, temporal_scale, aspect_ratio
        self.npred = npred
        self.max_ctx_t = max(1, int(self.t * max_temporal_keep))
        self.max_keep = max_keep
        self.counter = -1

    def _block_size(self, gen):
        r = torch.rand(1, generator=gen).item()
        t = max(1, int(self.t * (self.temporal_scale[0] + r * (self.temporal_scale[1] - self.temporal_scale[0]))))
        r = torch.rand(1, generator=gen).item()
        keep = int(self.h * self.w * (self.spatial_scale[0] + r * (self.spatial_scale[1] - self.spatial_scale[0])))
        r = torch.rand(1, generator=gen).item()
        ar = self.aspect_ratio[0] + r * (self.aspect_ratio[1] - self.aspect_ratio[0])
        bh = min(int(round(math.sqrt(keep * ar))), self.h)
        bw = min(int(round(math.sqrt(keep / ar))), self.w)
        return t, bh, bw

    def __call__(self, batch_size):
        self.counter += 1
        gen = torch.Generator()
        gen.manual_seed(self.counter)
        bt, bh, bw = self._block_size(gen)
        encs, preds = [], []
        min_e = min_p = self.t * self.h * self.w
        for _ in range(batch_size):
            while True:
                keep = np.ones((self.t, self.h, self.w), dtype=np.int32)
                for _ in range(self.npred):
                    top = int(torch.randint(0, self.h - bh + 1, (1,)))
                    left = int(torch.randint(0, self.w - bw + 1, (1,)))
                    start = int(torch.randint(0, self.t - bt + 1, (1,)))
                    blk = np.ones_like(keep)
                    blk[start:start + bt, top:top + bh, left:left + bw] = 0
                    if self.max_ctx_t < self.t:
                        blk[self.max_ctx_t:] = 0
                    keep *= blk
                flat = keep.reshape(-1)
                e = np.nonzero(flat)[0]
                if len(e) == 0:
                    continue
                p = np.nonzero(flat == 0)[0]
                min_e, min_p = min(min_e, len(e)), min(min_p, len(p))
                encs.append(e)
                preds.append(p)
                break
        if self.max_keep is not None:
            min_e = min(min_e, self.max_keep)
        enc = torch.from_numpy(np.stack([e[:min_e] for e in encs]).astype(np.int64))
        pred = torch.from_numpy(np.stack([p[:min_p] for p in preds]).astype(np.int64))
        return enc, pred


def make_mask_gens(cfgs_mask, crop_size, num_frames, patch, tubelet):
    """MaskCollator.__init__, multiblock3d.py:22-46."""
    return [MaskGenOracle(crop_size, num_frames, patch, tubelet, m["spatial_scale"], m["temporal_scale"],
                          m["aspect_ratio"], m["num_blocks"], m.get("max_temporal_keep", 1.0), m.get("max_keep"))
            for m in cfgs_mask]


# ------------------------------------------------------------------------------------------------------------
# bf16 STORAGE emulation (optional; `emu=True` on the functions below).  Still test infrastructure.
#
# The reference arithmetic above is fp32.  The HIP path computes the same expressions with fp32 accumulation but STORES
# every activation and every activation gradient in bf16 and multiplies against bf16 copies of the fp32 master weights.
# Against the fp32 oracle that rounding noise shows up as 1e-2 (features) ... 6e-2 (gradients of few-row tensors on the tiny
# models), which is why the small-model parity bounds of rounds 1-3 were loose enough for a 3x kernel regression to pass.  With
# `emu=True` the SAME functions round at the points where the HIP path stores (LayerNorm outputs, qkv with the
# soft-max scale applied to q before its rounding, the un-normalised soft-max probabilities, attention output, residual
# stream, pre-activation, GELU output, the saved GELU derivative, every gradient tensor of those) -- the rounding noise becomes
# common-mode and what is left is summation order, so the tiny-model bounds can be 10x tighter (tests/test_emu_parity_gpu.py).
# The default (`emu=False`) path is untouched and stays pinned bit for bit by the golden fixtures; the emulation adds no
# arithmetic of its own, only `.to(bfloat16)` round trips, so it inherits that pin.
# ------------------------------------------------------------------------------------------------------------
def _r(x):
    """Round to bf16 (RNE, as torch / the kernels' v_cvt_pk_bf16_f32) and back."""
    return x.to(torch.bfloat16).to(torch.float32)


class _RoundBoth(torch.autograd.Function):
    """A tensor the HIP path stores in bf16 together with its gradient: value and incoming gradient are rounded."""

    @staticmethod
    def forward(ctx, x):
        return _r(x)

    @staticmethod
    def backward(ctx, g):
        return _r(g)


class _RoundFwd(torch.autograd.Function):
    """bf16 shadow of an fp32 master weight (or any operand whose gradient stays fp32): value rounded, gradient untouched."""

    @staticmethod
    def forward(ctx, x):
        return _r(x)

    @staticmethod
    def backward(ctx, g):
        return g


class _RoundGrad(torch.autograd.Function):
    """Identity whose incoming gradient is rounded (the attention backward feeds bf16 dS to its dQ / dK products)."""

    @staticmethod
    def forward(ctx, x):
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return _r(g)


class _GeluStored(torch.autograd.Function):
    """fc1 epilogue + fc2-dgrad epilogue of the HIP path: g = bf16(gelu(u)) and, for the backward, the SAVED derivative
    bf16(gelu'(u)); du = dg * saved (dg = the fp32 accumulators of the fc2 dgrad GEMM, not rounded on its own)."""

    @staticmethod
    def forward(ctx, u):
        ud = u.double()
        phi = 0.5 * (1.0 + torch.erf(ud / math.sqrt(2.0)))
        ctx.save_for_backward(_r((phi + ud * torch.exp(-0.5 * ud * ud) / math.sqrt(2.0 * math.pi)).float()))
        return _r((ud * phi).float())

    @staticmethod
    def backward(ctx, dg):
        (d,) = ctx.saved_tensors
        return dg * d


def _q(x, emu):
    return _RoundBoth.apply(x) if emu else x


def _w(x, emu):
    return _RoundFwd.apply(x) if emu else x


LOG2E = 1.4426950408889634


# ------------------------------------------------------------------------------------------------------------
# model pieces -- src/models/utils/modules.py, patch_embed.py, vision_transformer.py, predictor.py
# ------------------------------------------------------------------------------------------------------------
def take_rows(x, idx):
    """apply_masks for one mask: src/masks/utils.py:17-19."""
    return torch.gather(x, 1, idx.unsqueeze(-1).expand(-1, -1, x.size(-1)))


# The EMA target encoder of the HIP path folds each LayerNorm into the Linear that consumes it (round 5, option ln_fold /
# VJ_LN_FOLD, default off): the normalised rows are never stored, the bf16 operand of the GEMM is W * gamma instead of W, and the
# bias absorbs W beta.  Same mathematics (LayerNorm(x) W^T + b), other rounding points; the emulation follows when this is True
# (the HIP default is OFF: tests that switch the fold on set this flag for their oracle run).
EMU_TARGET_LN_FOLD = False


def _folded_linear(x, W, b, gamma, beta, eps):
    """LayerNorm(x; gamma, beta) W^T + b with the storage points of vj_gemm_bf16_nt_lnfold: x_hat in fp32 (not stored), bf16(W * gamma),
    fp32 bias b + W beta.  No gradient flows here (the target encoder is frozen)."""
    xh = F.layer_norm(x, (x.shape[-1],), None, None, eps)
    return F.linear(xh, _r(W * gamma), b + W @ beta)


def _block_emu(x, w, pre, heads, eps, fold=False):
    """`block` with the HIP path's bf16 storage points (see the section comment above).  fold: the LayerNorms folded into the
    qkv / fc1 GEMMs (target encoder, EMU_TARGET_LN_FOLD)."""
    B, S, D = x.shape
    hd = D // heads
    if fold:
        qkv = _folded_linear(x, w[pre + "attn.qkv.weight"], w[pre + "attn.qkv.bias"], w[pre + "norm1.weight"], w[pre + "norm1.bias"], eps)
    else:
        y = _q(F.layer_norm(x, (D,), w[pre + "norm1.weight"], w[pre + "norm1.bias"], eps), True)
        qkv = F.linear(y, _w(w[pre + "attn.qkv.weight"], True), w[pre + "attn.qkv.bias"])
    q, k, v = qkv.reshape(B, S, 3, heads, hd).permute(2, 0, 3, 1, 4)
    q = _q(q * (hd ** -0.5 * LOG2E), True)      # ONE rounding of c*q (qkv GEMM epilogue 4), base-2 logits from here on
    k, v = _q(k, True), _q(v, True)
    s2 = _RoundGrad.apply(q @ k.transpose(-1, -2))                     # fp32 scores; dS reaches the dQ / dK products in bf16
    p = torch.exp2(s2 - s2.max(dim=-1, keepdim=True).values.detach())
    p = _w(p, True)                                                    # bf16 P feeds the P.V MFMA; its row sums are sums of those
    att = _q((p @ v) / p.sum(dim=-1, keepdim=True), True)
    att = att.transpose(1, 2).reshape(B, S, D)
    x = _q(x + F.linear(att, _w(w[pre + "attn.proj.weight"], True), w[pre + "attn.proj.bias"]), True)
    if fold:
        u = _q(_folded_linear(x, w[pre + "mlp.fc1.weight"], w[pre + "mlp.fc1.bias"], w[pre + "norm2.weight"], w[pre + "norm2.bias"], eps), True)
    else:
        y = _q(F.layer_norm(x, (D,), w[pre + "norm2.weight"], w[pre + "norm2.bias"], eps), True)
        u = _q(F.linear(y, _w(w[pre + "mlp.fc1.weight"], True), w[pre + "mlp.fc1.bias"]), True)
    g = _GeluStored.apply(u)
    return _q(x + F.linear(g, _w(w[pre + "mlp.fc2.weight"], True), w[pre + "mlp.fc2.bias"]), True)


def block(x, w, pre, heads, eps=1e-6, emu=False, fold=False):
    """Block.forward (modules.py:114-120) with Attention (61-78, SDPA branch) and MLP (30-36)."""
    if emu:
        return _block_emu(x, w, pre, heads, eps, fold=fold)
    B, S, D = x.shape
    y = F.layer_norm(x, (D,), w[pre + "norm1.weight"], w[pre + "norm1.bias"], eps)
    qkv = F.linear(y, w[pre + "attn.qkv.weight"], w[pre + "attn.qkv.bias"])
    q, k, v = qkv.reshape(B, S, 3, heads, D // heads).permute(2, 0, 3, 1, 4)
    att = F.scaled_dot_product_attention(q, k, v)  # modules.py:66-69 (use_sdpa is always True)
    att = att.transpose(1, 2).reshape(B, S, D)
    x = x + F.linear(att, w[pre + "attn.proj.weight"], w[pre + "attn.proj.bias"])
    y = F.layer_norm(x, (D,), w[pre + "norm2.weight"], w[pre + "norm2.bias"], eps)
    y = F.gelu(F.linear(y, w[pre + "mlp.fc1.weight"], w[pre + "mlp.fc1.bias"]))
    return x + F.linear(y, w[pre + "mlp.fc2.weight"], w[pre + "mlp.fc2.bias"])


def patchify(clips, tubelet, patch):
    """Non-overlapping Conv3d == unfold + GEMM (patch_embed.py:43-57): rows (t',h',w'), cols (c,dt,dh,dw)."""
    B, C, T, H, W = clips.shape
    u = clips.reshape(B, C, T // tubelet, tubelet, H // patch, patch, W // patch, patch)
    return u.permute(0, 2, 4, 6, 1, 3, 5, 7).reshape(B, -1, C * tubelet * patch * patch)


def encoder_forward(w, clips, cfg, masks=None, emu=False, round_out=True, fold=False):
    """VisionTransformer.forward (vision_transformer.py:159-195) for one mask (MultiMaskWrapper loops masks,
    multimask.py:17-26).  Returns [B, K or N, D] after the final norm.  emu: bf16 storage emulation; round_out=False
    leaves the final norm's output in fp32 (the HIP target path fuses it with the following F.layer_norm in fp32)."""
    D = cfg["embed_dim"]
    tok = patchify(clips, cfg["tubelet"], cfg["patch"])
    if emu:
        tok = _r(tok)
    x = _q(tok @ _w(w["patch_embed.proj.weight"], emu).reshape(D, -1).t() + w["patch_embed.proj.bias"], emu)
    x = _q(x + w["pos_embed"], emu)
    if masks is not None:
        x = take_rows(x, masks)
    for i in range(cfg["depth"]):
        x = block(x, w, f"blocks.{i}.", cfg["heads"], emu=emu, fold=fold and emu)
    return _q(F.layer_norm(x, (D,), w["norm.weight"], w["norm.bias"], 1e-6), emu and round_out)


def predictor_forward(w, z, idx_e, idx_p, mask_index, cfg, emu=False):
    """VisionTransformerPredictor.forward (predictor.py:174-239), mask-token branch."""
    Dp = cfg["pred_dim"]
    B, Ke = idx_e.shape
    x = _q(F.linear(z, _w(w["predictor_embed.weight"], emu), w["predictor_embed.bias"]), emu)
    pos = w["predictor_pos_embed"].expand(B, -1, -1)
    x = _q(x + take_rows(pos, idx_e), emu)
    tok = w[f"mask_tokens.{mask_index % cfg['num_mask_tokens']}"].reshape(1, 1, Dp)
    t = _q(tok + take_rows(pos, idx_p), emu)
    x = torch.cat([x, t], dim=1)
    for i in range(cfg["pred_depth"]):
        x = block(x, w, f"predictor_blocks.{i}.", cfg["heads"], emu=emu)
    x = _q(F.layer_norm(x[:, Ke:], (Dp,), w["predictor_norm.weight"], w["predictor_norm.bias"], 1e-6), emu) if emu else \
        F.layer_norm(x, (Dp,), w["predictor_norm.weight"], w["predictor_norm.bias"], 1e-6)[:, Ke:]
    return _q(F.linear(x, _w(w["predictor_proj.weight"], emu), w["predictor_proj.bias"]), emu)


# ------------------------------------------------------------------------------------------------------------
# schedules -- src/utils/schedulers.py:31-45,63-76 ; momentum generator app/vjepa/train.py:302-303
# ------------------------------------------------------------------------------------------------------------
def lr_at(step, warmup_steps, start_lr, ref_lr, final_lr, T_max_total):
    """WarmupCosineSchedule.step() return value at its `step`-th call (1-based)."""
    T_max = T_max_total - warmup_steps
    if step < warmup_steps:
        return start_lr + (step / max(1, warmup_steps)) * (ref_lr - start_lr)
    prog = (step - warmup_steps) / max(1, T_max)
    return max(final_lr, final_lr + (ref_lr - final_lr) * 0.5 * (1.0 + math.cos(math.pi * prog)))


def wd_at(step, ref_wd, final_wd, T_max):
    """CosineWDSchedule.step() at its `step`-th call."""
    wd = final_wd + (ref_wd - final_wd) * 0.5 * (1.0 + math.cos(math.pi * step / T_max))
    return max(final_wd, wd) if final_wd <= ref_wd else min(final_wd, wd)


def ema_at(i, ema0, ema1, ipe, epochs, ipe_scale):
    """i-th (0-based) value of the momentum generator, train.py:302-303."""
    return ema0 + i * (ema1 - ema0) / (ipe * epochs * ipe_scale)


def is_no_decay(name, p):
    """Param-group split of init_opt (app/vjepa/utils.py:173-191): bias or 1-D -> weight_decay 0."""
    return ("bias" in name) or (p.dim() == 1)


# ------------------------------------------------------------------------------------------------------------
# one optimisation step -- app/vjepa/train.py:414-487
# ------------------------------------------------------------------------------------------------------------
def loss_terms(z_list, h_list, loss_exp):
    """loss_fn / reg_fn, train.py:440-449,456-459."""
    loss = sum(torch.mean(torch.abs(z - h) ** loss_exp) / loss_exp for z, h in zip(z_list, h_list)) / len(z_list)
    pstd = sum(torch.sqrt(z.var(dim=1) + 0.0001) for z in z_list) / len(z_list)
    return loss, torch.mean(F.relu(1.0 - pstd))


def forward_all(enc_w, pred_w, tgt_w, clips, masks_enc, masks_pred, cfg, emu=False):
    """forward_target + forward_context, train.py:419-438."""
    D = cfg["embed_dim"]
    with torch.no_grad():
        h = encoder_forward(tgt_w, clips, cfg, emu=emu, round_out=False, fold=EMU_TARGET_LN_FOLD)
        h = F.layer_norm(h, (D,))  # eps 1e-5, no affine (train.py:426)
        h_list = [take_rows(h, mp) for mp in masks_pred]
    z_enc = [encoder_forward(enc_w, clips, cfg, me, emu=emu) for me in masks_enc]
    z_list = [predictor_forward(pred_w, ze, me, mp, i, cfg, emu=emu)
              for i, (ze, me, mp) in enumerate(zip(z_enc, masks_enc, masks_pred))]
    return h_list, z_enc, z_list


def adamw_update(p, g, state, lr, wd, beta1, beta2, eps, step):
    """torch.optim.AdamW single-tensor arithmetic (decoupled decay, bias-corrected), as invoked at train.py:475."""
    m, v = state
    p.mul_(1.0 - lr * wd)
    m.lerp_(g, 1.0 - beta1)
    v.mul_(beta2).addcmul_(g, g, value=1.0 - beta2)
    bc1 = 1.0 - beta1 ** step
    bc2 = 1.0 - beta2 ** step
    denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
    p.addcdiv_(m, denom, value=-lr / bc1)


def step_grads(state, clips, masks_enc, masks_pred, cfg, hp, emu=False):
    """Forward + loss + backward of train.py:419-464 on one batch: returns (checkpoints, grads) without touching
    `state`.  grads = {"enc": {name: g}, "pred": {name: g}}.  emu: bf16 storage emulation (section comment above)."""
    enc_w = {k: v.detach().clone().requires_grad_(k != "pos_embed") for k, v in state["enc"].items()}
    pred_w = {k: v.detach().clone().requires_grad_(k != "predictor_pos_embed") for k, v in state["pred"].items()}
    h_list, z_enc, z_list = forward_all(enc_w, pred_w, state["tgt"], clips, masks_enc, masks_pred, cfg, emu=emu)
    loss_jepa, loss_reg = loss_terms(z_list, h_list, hp["loss_exp"])
    loss = loss_jepa + hp["reg_coeff"] * loss_reg
    loss.backward()
    grads = {"enc": {k: v.grad for k, v in enc_w.items() if v.grad is not None},
             "pred": {k: v.grad for k, v in pred_w.items() if v.grad is not None}}
    out = {"loss": float(loss.detach()), "loss_jepa": float(loss_jepa.detach()), "loss_reg": float(loss_reg.detach()),
           "h": h_list, "z_enc": [z.detach() for z in z_enc], "z": [z.detach() for z in z_list]}
    return out, grads


def apply_update(state, grads, hp, step, clip_now=False):
    """clip_grad_norm_ (train.py:466-470, per module, only when active) + AdamW (train.py:471-475 via
    app/vjepa/utils.py:173-194) + EMA (train.py:483-487) on `state`, in place.  Returns (lr, wd, ema, norms)."""
    lr = lr_at(step, int(hp["warmup"] * hp["ipe"]), hp["start_lr"], hp["lr"], hp["final_lr"],
               int(hp["ipe_scale"] * hp["epochs"] * hp["ipe"]))
    wd = wd_at(step, hp["wd"], hp["final_wd"], int(hp["ipe_scale"] * hp["epochs"] * hp["ipe"]))
    norms = (0.0, 0.0)
    if clip_now and hp.get("clip_grad") is not None:
        # torch.nn.utils.clip_grad_norm_ operates on .grad of parameters: wrap the gradient tensors accordingly
        ns = []
        for grp in ("enc", "pred"):
            ps = []
            for g in grads[grp].values():
                p = torch.nn.Parameter(torch.zeros_like(g))
                p.grad = g            # clipped in place
                ps.append(p)
            ns.append(float(torch.nn.utils.clip_grad_norm_(ps, hp["clip_grad"])))
        norms = tuple(ns)
    with torch.no_grad():
        for grp in ("enc", "pred"):
            for name, p in state[grp].items():
                g = grads[grp].get(name)
                if g is None:
                    continue
                st = state["opt"].setdefault(grp + "." + name, (torch.zeros_like(p), torch.zeros_like(p)))
                adamw_update(p, g, st, lr, 0.0 if is_no_decay(name, p) else wd, hp["betas"][0], hp["betas"][1],
                             hp["eps"], step)
        m = ema_at(step - 1, hp["ema"][0], hp["ema"][1], hp["ipe"], hp["epochs"], hp["ipe_scale"])
        for name, pk in state["tgt"].items():  # train.py:486-487 (frozen pos_embed rides along)
            pk.mul_(m).add_((1.0 - m) * state["enc"][name])
    return lr, wd, m, norms


def train_step(state, clips, masks_enc, masks_pred, cfg, hp, step, clip_now=False):
    """One full step on `state` = dict(enc=, pred=, tgt=, opt={name:(m,v)}) of fp32 CPU tensors (updated in
    place).  `step` is 1-based.  Returns a dict of checkpoints for parity tests."""
    out, grads = step_grads(state, clips, masks_enc, masks_pred, cfg, hp)
    raw = {grp: {k: v.clone() for k, v in gs.items()} for grp, gs in grads.items()} if clip_now else grads
    lr, wd, m, norms = apply_update(state, grads, hp, step, clip_now)
    out.update({"lr": lr, "wd": wd, "ema": m, "grads": raw, "grad_norms": norms})
    return out


def train_step_dp(state, rank_batches, cfg, hp, step, clip_now=False):
    """The same step under DistributedDataParallel (train.py:295-297): every rank runs forward/backward on its own
    (clips, masks_enc, masks_pred) and the gradients are AVERAGED over ranks before clip / AdamW / EMA; the loss each
    rank logs is its own.  rank_batches: list of (clips, masks_enc, masks_pred), one per rank."""
    outs, acc = [], None
    for clips, me, mp in rank_batches:
        o, g = step_grads(state, clips, me, mp, cfg, hp)
        outs.append(o)
        if acc is None:
            acc = {grp: {k: v.clone() for k, v in gs.items()} for grp, gs in g.items()}
        else:
            for grp in acc:
                for k in acc[grp]:
                    acc[grp][k] += g[grp][k]
    W = len(rank_batches)
    for grp in acc:
        for k in acc[grp]:
            acc[grp][k] /= W
    raw = {grp: {k: v.clone() for k, v in gs.items()} for grp, gs in acc.items()}
    lr, wd, m, norms = apply_update(state, acc, hp, step, clip_now)
    return {"ranks": outs, "lr": lr, "wd": wd, "ema": m, "grads": raw, "grad_norms": norms}


# ------------------------------------------------------------------------------------------------------------
# FLOP model of SURVEY.md section 8(d): matmul FLOPs only, bwd = 2x fwd
# ------------------------------------------------------------------------------------------------------------
def step_flops(cfg, B, Ke_list, Kp_list):
    D, Dp = cfg["embed_dim"], cfg["pred_dim"]
    kdim = 3 * cfg["tubelet"] * cfg["patch"] ** 2
    N = cfg["num_patches"]

    def blk(s, d):
        return 24 * s * d * d + 4 * s * s * d

    f_tgt = cfg["depth"] * blk(N, D) + 2 * N * kdim * D
    f_ctx = sum(cfg["depth"] * blk(ke, D) + 2 * ke * kdim * D for ke in Ke_list)
    f_pred = sum(cfg["pred_depth"] * blk(ke + kp, Dp) + 2 * ke * D * Dp + 2 * kp * Dp * D
                 for ke, kp in zip(Ke_list, Kp_list))
    return B * (f_tgt + 3 * (f_ctx + f_pred))
