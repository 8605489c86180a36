"""The V-JEPA pretraining step on MI355X: target forward, context+predictor forward/backward, latent loss,
gradient all-reduce, fused AdamW + EMA + bf16 re-cast -- the arithmetic of the reference closure
app/vjepa/train.py:414-487, executed by hand-written gfx950 kernels with no autograd graph.

    trainer = Trainer(encoder, predictor, target_encoder, ...)      # modules from init_video_model
    out = trainer.train_step(clips, masks_enc, masks_pred, lr=..., wd=..., ema=...)
    out.loss / out.loss_jepa / out.loss_reg                          # lazily synchronising floats

Divergences from the reference that are deliberate: GradScaler is not emulated (bf16 has fp32's
exponent range; the scaler's only observable effect is skipping a step on non-finite grads, which we keep as a
device-side check when `check_finite=True`); the EMA skips the frozen pos_embed (m*x + (1-m)*x == x up to one
rounding); both masks run through one fused chain.
"""
import math
from dataclasses import dataclass

import torch

from ..hip import ops
from . import dp
from .layers import encoder_backward, encoder_forward, predictor_backward, predictor_forward, side_stream
from .weights import ParamArena, encoder_views, is_no_decay, predictor_views


class _TargetArena:
    """EMA target weights: fp32 + bf16 arenas laid out exactly like the encoder range of the trainer arena."""

    def __init__(self, src: ParamArena, lo, hi, prefix, target_named_params, device):
        self.device = device
        self.slots = {n: s for n, s in src.slots.items() if n.startswith(prefix)}
        self.lo, self.hi = lo, hi
        self.P = torch.zeros(hi - lo, dtype=torch.float32, device=device)
        self.Pb = torch.zeros(hi - lo, dtype=torch.bfloat16, device=device)
        self.wT = {}
        self.frozen = {}
        with torch.no_grad():
            for name, p in target_named_params:
                s = self.slots[name]
                v = self.P[s.off - lo:s.off - lo + s.numel]
                v.copy_(p.data.reshape(-1).to(device=device, dtype=torch.float32))
                p.data = v.view(s.shape)
        ops.cast_bf16(self.P, self.Pb)

    def f32(self, name):
        s = self.slots[name]
        return self.P[s.off - self.lo:s.off - self.lo + s.numel].view(s.shape)

    def bf16(self, name):
        s = self.slots[name]
        return self.Pb[s.off - self.lo:s.off - self.lo + s.numel].view(s.shape)

    def refresh_bf16(self):
        ops.cast_bf16(self.P, self.Pb)


class StepOutput:
    """Losses stay on the device until read (one host sync for all of them)."""

    def __init__(self, buf, reg_coeff, lr, wd, ema, grad_norms):
        self._buf, self._reg_coeff = buf, reg_coeff
        self.lr, self.wd, self.ema = lr, wd, ema
        self.grad_norms = grad_norms
        self._host = None

    def _fetch(self):
        if self._host is None:
            self._host = self._buf.tolist()
        return self._host

    @property
    def loss_jepa(self):
        return self._fetch()[0]

    @property
    def loss_reg(self):
        return self._fetch()[1]

    @property
    def loss(self):
        return self.loss_jepa + self._reg_coeff * self.loss_reg


def _strip(name):
    return name[len("backbone."):] if name.startswith("backbone.") else name


class Trainer:
    def __init__(self, encoder, predictor, target_encoder, loss_exp=1.0, reg_coeff=0.0, betas=(0.9, 0.999),
                 eps=1e-8, clip_grad=None, device=None, world_size=1, overlap_comm=True, check_finite=False):
        self.encoder, self.predictor, self.target_encoder = encoder, predictor, target_encoder
        self.vit, self.pred, self.tvit = encoder.backbone, predictor.backbone, target_encoder.backbone
        self.device = torch.device(device) if device is not None else next(encoder.parameters()).device
        if self.device.type != "cuda":
            raise ValueError("jepa_amd.Trainer needs a GPU device: the step runs in libvjepa_hip.so only")
        self.loss_exp, self.reg_coeff = float(loss_exp), float(reg_coeff)
        self.betas, self.eps, self.clip_grad = tuple(betas), float(eps), clip_grad
        self.check_finite = check_finite
        self.world_size = world_size
        dev = self.device

        enc_named = [("enc." + _strip(n), n, p) for n, p in encoder.named_parameters()]
        pred_named = [("pred." + _strip(n), n, p) for n, p in predictor.named_parameters()]
        # the four AdamW groups of init_opt (reference app/vjepa/utils.py:173-191)
        g_enc_d = [(a, p) for a, n, p in enc_named if p.requires_grad and not is_no_decay(n, p)]
        g_enc_n = [(a, p) for a, n, p in enc_named if p.requires_grad and is_no_decay(n, p)]
        g_pred_d = [(a, p) for a, n, p in pred_named if p.requires_grad and not is_no_decay(n, p)]
        g_pred_n = [(a, p) for a, n, p in pred_named if p.requires_grad and is_no_decay(n, p)]
        self.arena = ParamArena([g_enc_d, g_enc_n, g_pred_d, g_pred_n], dev)
        for a, n, p in enc_named + pred_named:
            if not p.requires_grad:
                p.data = p.data.to(device=dev, dtype=torch.float32).contiguous()
                self.arena.frozen[a] = p.data
        lin = [a + ".weight" for a in
               ["enc." + n for n in self.vit._hip_linear_names()] + ["pred." + n for n in self.pred._hip_linear_names()]]
        self.arena.make_transposed(lin)  # dgrad operands W^T (the patch embed needs none: pixels get no grad)
        self.vit._hip_attach(self.arena, "enc.")
        self.pred._hip_attach(self.arena, "pred.")
        # EMA target: same layout as the encoder range
        enc_lo, enc_hi = self.arena.group_ranges[0][0], self.arena.group_ranges[1][1]
        tgt_named = [("enc." + _strip(n), p) for n, p in target_encoder.named_parameters()]
        self.tarena = _TargetArena(self.arena, enc_lo, enc_hi, "enc.",
                                   [(a, p) for a, p in tgt_named if a in self.arena.slots], dev)
        for a, p in tgt_named:
            if a not in self.arena.slots:
                p.data = p.data.to(device=dev, dtype=torch.float32).contiguous()
                self.tarena.frozen[a] = p.data
        self.tvit._hip_attach(self.tarena, "enc.")
        self.ew = encoder_views(self.arena, "enc.", self.vit,
                                self.arena.frozen["enc.pos_embed"].reshape(self.vit.num_patches, -1), train=True)
        self.pw = predictor_views(self.arena, "pred.", self.pred,
                                  self.arena.frozen["pred.predictor_pos_embed"].reshape(self.pred.num_patches, -1),
                                  train=True)
        self.tw = encoder_views(self.tarena, "enc.", self.tvit,
                                self.tarena.frozen["enc.pos_embed"].reshape(self.tvit.num_patches, -1), train=False)
        # optimizer-facing view (schedulers write lr / weight_decay into these dicts, like torch param_groups);
        # order = the reference's: [enc decayed, pred decayed, enc no-decay, pred no-decay]
        self.param_groups = [
            {"params": [p for _, p in g_enc_d], "lr": 0.0, "weight_decay": 0.0, "_range": 0},
            {"params": [p for _, p in g_pred_d], "lr": 0.0, "weight_decay": 0.0, "_range": 2},
            {"params": [p for _, p in g_enc_n], "lr": 0.0, "weight_decay": 0, "WD_exclude": True, "_range": 1},
            {"params": [p for _, p in g_pred_n], "lr": 0.0, "weight_decay": 0, "WD_exclude": True, "_range": 3},
        ]
        self.opt_step = 0
        self._stat = torch.zeros(8, dtype=torch.float32, device=dev)   # [loss_jepa, loss_reg, -, -, sq_enc, bad, sq_pred, bad]
        self.reducer = dp.GradReducer(self.arena, self.vit, self.pred, world_size, overlap=overlap_comm)

    # ------------------------------------------------------------------------------------------------ forward
    @torch.no_grad()
    def forward_target(self, clips, masks_pred):
        """h_i = apply_masks(F.layer_norm(target_encoder(clips)), masks_pred)  (train.py:419-429), fp32."""
        B = clips.shape[0]
        x, _, _ = encoder_forward(self.tw, clips, None, save=False, final_norm=False)
        N = self.tvit.num_patches
        return [ops.target_rows(x, self.tw.norm.g, self.tw.norm.b, mp, B, N, 1e-6, 1e-5) for mp in masks_pred]

    @torch.no_grad()
    def train_step(self, clips, masks_enc, masks_pred, lr, wd, ema, clip_now=False):
        """One optimisation step.  clips fp32 [B,3,T,H,W]; masks_*: lists of int64 [B,K] (device tensors)."""
        assert len(masks_enc) == len(masks_pred), 'Currently require num encoder masks = num predictor masks'
        B = clips.shape[0]
        D = self.vit.embed_dim
        n_masks = len(masks_pred)
        mark = self._phase_mark
        mark('start')
        # ---- forward: the EMA target branch is independent of the context branch until the loss, so it runs on
        #      the side stream concurrently (fills the tails of each other's kernels)
        side = side_stream(self.device)
        if side.enabled:
            side.fork(clips, *masks_pred)
            with torch.cuda.stream(side.stream):
                h = self.forward_target(clips, masks_pred)
        else:
            h = self.forward_target(clips, masks_pred)
        z, segs, saved_e = encoder_forward(self.ew, clips, masks_enc, save=True)
        zhat, tsegs, saved_p = predictor_forward(self.pw, z, segs, masks_enc, masks_pred, save=True)
        mark('context+predictor forward (main stream)')
        if side.enabled:
            side.join()
            for t in h:
                t.record_stream(torch.cuda.current_stream())
        # ---- loss (train.py:440-459) and its gradient; |grad| carried as {+-s_i} in bf16, the common factor
        #      alpha = 1/(numel_min * n_masks) is applied in fp32 where parameter gradients are written
        numels = [t.rows * D for t in tsegs]
        nmin = min(numels)
        alpha = 1.0 / (nmin * n_masks)
        dzhat = torch.empty_like(zhat)
        pstd = torch.empty((B, D), dtype=torch.float32, device=self.device)
        with_reg = self.reg_coeff != 0.0
        stats = [torch.empty((B, D, 2), dtype=torch.float32, device=self.device) if with_reg else None for _ in tsegs]
        for i, t in enumerate(tsegs):
            zi = zhat[t.row0:t.row0 + t.rows]
            ops.latent_loss(zi, h[i], self._stat[0:1], p=self.loss_exp, out_scale=1.0 / (numels[i] * n_masks),
                            accumulate=i > 0, dz=dzhat[t.row0:t.row0 + t.rows], gscale=nmin / numels[i])
            ops.token_pstd(zi, pstd, B, t.S, D, accumulate=i > 0, stats=stats[i])
        ops.reg_finish(pstd, n_masks, self._stat[1:2])
        if with_reg:   # gradient of reg_coeff * mean(relu(1 - pstd)), expressed in the 1/alpha units dzhat carries
            coef = self.reg_coeff / (B * D * n_masks) / alpha
            for i, t in enumerate(tsegs):
                ops.reg_grad(zhat[t.row0:t.row0 + t.rows], pstd, stats[i], dzhat[t.row0:t.row0 + t.rows], B, t.S, D,
                             n_masks, coef)
        mark('target forward joined + loss')
        # ---- backward (predictor first, then encoder layers 23..0); gradient buckets go out as layers finish
        side = side_stream(self.device)
        self.reducer.begin(side.stream if side.enabled else None)
        hook = self.reducer.layer_done if self.reducer.enabled else None
        dz = predictor_backward(dzhat, saved_p, self.pw, segs, alpha, on_layer_done=hook)
        mark('predictor backward (main stream)')
        encoder_backward(dz, saved_e, self.ew, segs, alpha, on_layer_done=hook)
        mark('encoder backward (main stream)')
        self.reducer.finish()
        # ---- clip / AdamW / EMA / bf16 re-cast (train.py:461-487)
        norms = self.optimizer_step(lr, wd, ema, clip_now)
        mark('wgrad stream joined + AdamW/EMA')
        return StepOutput(self._stat[:2].clone(), self.reg_coeff, lr, wd, ema, norms)

    def _phase_mark(self, name):
        """Phase timestamps on the main stream (tools / bench diagnostics): set `self.phase_events = []` to collect
        (name, event) pairs for the following steps; None (default) costs nothing."""
        ev = getattr(self, 'phase_events', None)
        if ev is not None:
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            ev.append((name, e))

    # ------------------------------------------------------------------------------------------------ update
    def _grad_sqnorms(self):
        A = self.arena
        (e0, _), (_, e1) = A.group_ranges[0], A.group_ranges[1]
        (p0, _), (_, p1) = A.group_ranges[2], A.group_ranges[3]
        ops.sqnorm(A.G[e0:e1], self._stat[4:6])
        ops.sqnorm(A.G[p0:p1], self._stat[6:8])
        return self._stat[4:8].tolist()

    def optimizer_step(self, lr, wd, ema, clip_now=False):
        A = self.arena
        inv_world = 1.0 / self.world_size
        scale_enc = scale_pred = inv_world
        norms = (0.0, 0.0)
        if (clip_now and self.clip_grad is not None) or self.check_finite:
            sq_e, bad_e, sq_p, bad_p = self._grad_sqnorms()   # one host sync, like clip_grad_norm_'s float()
            if self.check_finite and (bad_e + bad_p) > 0:
                return (float("nan"), float("nan"))            # GradScaler semantics: skip the step on inf/nan
            ne, npd = math.sqrt(sq_e) * inv_world, math.sqrt(sq_p) * inv_world
            norms = (ne, npd)
            if clip_now and self.clip_grad is not None:        # torch.nn.utils.clip_grad_norm_ coefficient
                scale_enc *= min(1.0, self.clip_grad / (ne + 1e-6))
                scale_pred *= min(1.0, self.clip_grad / (npd + 1e-6))
        self.opt_step += 1
        b1, b2 = self.betas
        T = self.tarena
        for gi, (lo, hi) in enumerate(A.group_ranges):
            if hi == lo:
                continue
            is_enc = gi < 2
            decay = wd if gi in (0, 2) else 0.0
            tgt = T.P[lo - T.lo:hi - T.lo] if is_enc else None
            tgtb = T.Pb[lo - T.lo:hi - T.lo] if is_enc else None
            ops.adamw_ema(A.P[lo:hi], A.G[lo:hi], A.M1[lo:hi], A.M2[lo:hi], A.Pb[lo:hi], tgt, tgtb, lr, decay, b1, b2,
                          self.eps, self.opt_step, scale_enc if is_enc else scale_pred, ema)
        A.refresh_transposed()
        for g in self.param_groups:
            g["lr"] = lr
            if not g.get("WD_exclude", False):
                g["weight_decay"] = wd
        return norms

    def zero_grad(self, set_to_none=False):
        """Gradients are fully overwritten by every backward (beta = 0 wgrads); kept for API compatibility."""
        return None

    # ------------------------------------------------------------------------------------------------ checkpoints
    def state_dict(self):
        """torch.optim.AdamW-compatible optimizer state (reference checkpoint key 'opt', train.py:331-342)."""
        state, groups, k = {}, [], 0
        b1, b2 = self.betas
        for g in self.param_groups:
            ids = []
            for p in g["params"]:
                s = next(sl for sl in self.arena.slots.values() if sl.param is p)
                state[k] = {"step": torch.tensor(float(self.opt_step)),
                            "exp_avg": self.arena.M1[s.off:s.off + s.numel].view(s.shape).clone(),
                            "exp_avg_sq": self.arena.M2[s.off:s.off + s.numel].view(s.shape).clone()}
                ids.append(k)
                k += 1
            gd = {kk: v for kk, v in g.items() if kk not in ("params", "_range")}
            gd.update({"params": ids, "betas": (b1, b2), "eps": self.eps, "amsgrad": False, "maximize": False,
                       "foreach": None, "capturable": False, "differentiable": False, "fused": None})
            groups.append(gd)
        return {"state": state, "param_groups": groups}

    def load_state_dict(self, sd):
        k = 0
        for g, gs in zip(self.param_groups, sd["param_groups"]):
            for p in g["params"]:
                s = next(sl for sl in self.arena.slots.values() if sl.param is p)
                st = sd["state"].get(k)
                if st is not None:
                    self.arena.M1[s.off:s.off + s.numel].copy_(st["exp_avg"].reshape(-1))
                    self.arena.M2[s.off:s.off + s.numel].copy_(st["exp_avg_sq"].reshape(-1))
                    self.opt_step = int(float(st["step"]))
                k += 1
            g["lr"] = gs.get("lr", g["lr"])
            g["weight_decay"] = gs.get("weight_decay", g["weight_decay"])

    def sync_shadows(self):
        """Call after writing parameters from outside (load_state_dict): refresh bf16 / transposed shadows."""
        self.arena.refresh_bf16()
        self.arena.refresh_transposed()
        self.tarena.refresh_bf16()
