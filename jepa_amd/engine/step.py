"""The V-JEPA pretraining step on MI355X: target forward, context+predictor forward/backward, latent loss,
gradient all-reduce, fused AdamW + EMA + bf16 re-cast -- the arithmetic of the reference closure
app/vjepa/train.py:414-487, executed by hand-written gfx950 kernels with no autograd graph.

    trainer = Trainer(encoder, predictor, target_encoder, ...)      # modules from init_video_model
    out = trainer.train_step(clips, masks_enc, masks_pred, lr=..., wd=..., ema=...)
    out.loss / out.loss_jepa / out.loss_reg                          # lazily synchronising floats

Divergences from the reference that are deliberate: GradScaler's loss scaling is not emulated (bf16
has fp32's exponent range); its one observable effect -- skipping the optimizer step when a gradient is non-finite --
is kept, always on, as a device-side flag that the fused AdamW kernel reads (no host sync); the EMA skips the frozen
pos_embed (m*x + (1-m)*x == x up to one rounding); both masks run through one fused chain; a batch larger than
`micro_batch` is processed in micro-batches with gradient accumulation (same sums, the collator still sees the whole
batch).
"""
import math
from dataclasses import dataclass

import torch

from ..hip import ops
import weakref

from . import chain, dp, optstate
from .layers import encoder_backward, encoder_forward, low_priority_stream, predictor_backward, predictor_forward, side_stream
from . import weights as _weights
from .weights import ParamArena, bump_generation, encoder_views, is_no_decay, predictor_views


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

    # ---- LayerNorms folded into the qkv / fc1 GEMMs of the target forward (vj_blocks_fwd_lnfold)
    def make_folds(self, depth):
        """One FoldW per block, refreshed in place from the fp32 target weights after every EMA update."""
        from .weights import FoldW
        self.folds = []
        for i in range(depth):
            wq, w1 = self.f32(f"enc.blocks.{i}.attn.qkv.weight"), self.f32(f"enc.blocks.{i}.mlp.fc1.weight")
            mk = lambda n: torch.empty(n, dtype=torch.float32, device=self.device)
            self.folds.append(FoldW(torch.empty_like(wq, dtype=torch.bfloat16), mk(wq.shape[0]), mk(wq.shape[0]),
                                    torch.empty_like(w1, dtype=torch.bfloat16), mk(w1.shape[0]), mk(w1.shape[0])))
        self.refresh_folds(0, depth)
        return self.folds

    def refresh_folds(self, b0, b1):
        """Wf = bf16(W * gamma), c = row sums of Wf, b' = b + W beta for blocks [b0, b1) -- two launches per block, on the
        current stream (the update stream when the update is deferred: before the range's gate event)."""
        if getattr(self, "folds", None) is None:
            return
        for i in range(b0, b1):
            f, pre = self.folds[i], f"enc.blocks.{i}."
            ops.ln_fold_weights(self.f32(pre + "attn.qkv.weight"), self.f32(pre + "attn.qkv.bias"), self.f32(pre + "norm1.weight"),
                                self.f32(pre + "norm1.bias"), f.w_qkv, f.c_qkv, f.b_qkv)
            ops.ln_fold_weights(self.f32(pre + "mlp.fc1.weight"), self.f32(pre + "mlp.fc1.bias"), self.f32(pre + "norm2.weight"),
                                self.f32(pre + "norm2.bias"), f.w_fc1, f.c_fc1, f.b_fc1)


class StepOutput:
    """Losses and gradient statistics stay on the device until read (one host sync for all of them)."""

    def __init__(self, buf, reg_coeff, lr, wd, ema, clipped, inv_world):
        self._buf, self._reg_coeff = buf, reg_coeff
        self.lr, self.wd, self.ema = lr, wd, ema
        self._clipped, self._inv_world = clipped, inv_world
        self._host = None

    def _fetch(self):
        if self._host is None:
            self._host = self._buf.tolist()   # [loss_jepa, loss_reg, -, -, sumsq_enc, bad_enc, sumsq_pred, bad_pred]
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

    @property
    def raw_grad_norms(self):
        """(encoder, predictor) L2 norms of the averaged gradients of this step (before clipping)."""
        h = self._fetch()
        return (math.sqrt(h[4]) * self._inv_world, math.sqrt(h[6]) * self._inv_world)

    @property
    def grad_norms(self):
        """What the reference logs (train.py:466-470): the clip_grad_norm_ results, 0 while clipping is inactive."""
        return self.raw_grad_norms if self._clipped else (0.0, 0.0)

    @property
    def skipped(self):
        """True when a non-finite gradient made the optimizer skip this step (GradScaler.step semantics)."""
        h = self._fetch()
        return (h[5] + h[7]) > 0


import os as _os
_OVERLAP_FWD = _os.environ.get("VJ_OVERLAP_FWD", "1") != "0"   # diagnostics: 0 = target forward on the main stream
# Fold the target encoder's LayerNorms into its qkv / fc1 GEMMs (vj_blocks_fwd_lnfold).  Built, parity-tested and measured in round 5:
# it removes the 48 LayerNorm launches of the target forward and their output traffic, but the heavier GEMM epilogue takes the gain
# back (profiles/r05_ln_fold.md), so it is opt-in (VJ_LN_FOLD=1 or Trainer.set_ln_fold(True)).
_LN_FOLD = _os.environ.get("VJ_LN_FOLD", "0") == "1"
_UPD_LOW_PRIO = _os.environ.get("VJ_UPD_LOW_PRIO", "0") == "1"   # the deferred update's stream at the device's lowest priority (A/B)
# GEMM kernel selection of the EMA target encoder's forward (vj_blocks_fwd gemm_flags: low 16 bits = flags, bits 16-23 = first
# block they apply to); 0 = automatic everywhere
_TGT_GEMM_FLAGS = int(_os.environ.get("VJ_TGT_GEMM_FLAGS", "0"), 0)
# the same for the predictor trunk (A/B measurements: tools/abab.py pred_flags / pred_dgrad_flags; 0 = the automatic selection):
# forward through vj_blocks_fwd's gemm_flags, backward by setting option gemm_dgrad_flags around the predictor's backward chain only
_PRED_GEMM_FLAGS = int(_os.environ.get("VJ_PRED_GEMM_FLAGS", "0"), 0)
_PRED_DGRAD_FLAGS = int(_os.environ.get("VJ_PRED_DGRAD_FLAGS", "0"), 0)


def _strip(name):
    return name[len("backbone."):] if name.startswith("backbone.") else name


class Trainer:
    def __init__(self, encoder, predictor, target_encoder, loss_exp=1.0, reg_coeff=0.0, betas=(0.9, 0.999),
                 eps=1e-8, clip_grad=None, device=None, world_size=1, overlap_comm=True, check_finite=True,
                 micro_batch=None, overlap_update=False):
        self.encoder, self.predictor, self.target_encoder = encoder, predictor, target_encoder
        self.vit, self.pred, self.tvit = encoder.backbone, predictor.backbone, target_encoder.backbone
        self.device = torch.device(device) if device is not None else next(encoder.parameters()).device
        if self.device.type != "cuda":
            raise ValueError("jepa_amd.Trainer needs a GPU device: the step runs in libvjepa_hip.so only")
        if self.device.index is None:   # 'cuda' and 'cuda:0' must name the same streams / workspaces everywhere
            self.device = torch.device("cuda", torch.cuda.current_device())
        self.micro_batch = micro_batch
        self._ws = f"trainer{id(self)}:"   # workspace namespace: two Trainers in one process never share activations
        weakref.finalize(self, chain.Workspace.release, self._ws)
        self.loss_exp, self.reg_coeff = float(loss_exp), float(reg_coeff)
        self.betas, self.eps, self.clip_grad = tuple(betas), float(eps), clip_grad
        self.check_finite = True   # always on and device-side (the argument is kept for API compatibility)
        self.world_size = world_size
        dev = self.device

        enc_named = [("enc." + _strip(n), n, p) for n, p in encoder.named_parameters()]
        pred_named = [("pred." + _strip(n), n, p) for n, p in predictor.named_parameters()]
        self._enc_named, self._pred_named = enc_named, pred_named
        self._gs_desc = self._gs_meta = self._arena_stats = None
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
        # LayerNorms of the EMA target encoder folded into its qkv / fc1 GEMMs (no LayerNorm launch, no LayerNorm output in HBM on
        # the target path).  `ln_fold_target` may be flipped between steps (tools/abab.py).
        self._target_folds = None
        self.ln_fold_target = False
        if _LN_FOLD:
            self.set_ln_fold(True)
        # optimizer-facing view (schedulers write lr / weight_decay into these dicts, like torch param_groups);
        # order = the reference's: [enc decayed, pred decayed, enc no-decay, pred no-decay].  Like init_opt
        # (app/vjepa/utils.py:173-191) the groups are built from ALL named_parameters, so the frozen pos_embed /
        # predictor_pos_embed sit in groups 0 / 1 as stateless members: parameter ids in state_dict() then line up
        # with a torch.optim.AdamW built by the reference.
        def grp(named, nodecay):
            return [p for _, n, p in named if is_no_decay(n, p) == nodecay]
        self.param_groups = [
            {"params": grp(enc_named, False), "lr": 0.0, "weight_decay": 0.0, "_range": 0},
            {"params": grp(pred_named, False), "lr": 0.0, "weight_decay": 0.0, "_range": 2},
            {"params": grp(enc_named, True), "lr": 0.0, "weight_decay": 0, "WD_exclude": True, "_range": 1},
            {"params": grp(pred_named, True), "lr": 0.0, "weight_decay": 0, "WD_exclude": True, "_range": 3},
        ]
        self._slot_of = {id(sl.param): sl for sl in self.arena.slots.values()}
        self._step_dev = torch.zeros(1, dtype=torch.float32, device=dev)   # Adam step count t (advanced on the device)
        self._stat = torch.zeros(8, dtype=torch.float32, device=dev)   # [loss_jepa, loss_reg, -, -, sq_enc, bad, sq_pred, bad]
        self.reducer = dp.GradReducer(self.arena, self.vit, self.pred, world_size, overlap=overlap_comm)
        # overlap_update: train_step returns with the fused AdamW / EMA update only ENQUEUED, on its own stream, range by range in
        # the order the next step's forward consumes the weights; the next train_step waits per range instead of for the whole
        # update (the update is 12.7 GB of HBM traffic that used to run with the chip otherwise empty).  Same kernels over the
        # same elements: results are bit-identical.  Off by default because anything that reads parameters with plain torch
        # operators right after train_step must then call sync_update() first (everything inside this package does).
        self.overlap_update = bool(overlap_update)
        self._upd_stream = self._make_update_stream() if self.overlap_update else None
        self.reducer.extra_streams = [self._upd_stream]   # the communication stream is picked against this one too
        self.reducer.prepare()   # communication stream + where torch.distributed's collectives land: now, not inside the first backward
        self._gates = None          # {'enc': [(first_block, event)], 'pred': [(0, event)], 'done': event} of the pending update
        self._plan = self._update_plan()

    # ------------------------------------------------------------------------------------------------ forward
    @torch.no_grad()
    def forward_target(self, clips, masks_pred):
        """h_i = apply_masks(F.layer_norm(target_encoder(clips)), masks_pred)  (train.py:419-429), fp32."""
        B = clips.shape[0]
        self.tw.folds = self._target_folds if (self.ln_fold_target and self._target_folds is not None) else None
        x, _, _ = encoder_forward(self.tw, clips, None, save=False, final_norm=False, ws_tag=self._ws + "tgt",
                                  gemm_flags=_TGT_GEMM_FLAGS, gates=self._gate("enc"))
        N = self.tvit.num_patches
        return [ops.target_rows(x, self.tw.norm.g, self.tw.norm.b, mp, B, N, 1e-6, 1e-5) for mp in masks_pred]

    @torch.no_grad()
    def train_step(self, clips, masks_enc, masks_pred, lr, wd, ema, clip_now=False):
        """One optimisation step.  clips fp32 [B,3,T,H,W]; masks_*: lists of int64 [B,K] (device tensors).
        With `micro_batch` set and B larger, the batch is walked in micro-batches: gradients accumulate in the arena
        (beta = 1 weight-gradient epilogues), losses accumulate on the device, one optimizer step at the end."""
        assert len(masks_enc) == len(masks_pred), 'Currently require num encoder masks = num predictor masks'
        B = clips.shape[0]
        D = self.vit.embed_dim
        n_masks = len(masks_pred)
        mark = self._phase_mark
        mark('start')
        self._arena_stats = None
        mb = B if not self.micro_batch else min(int(self.micro_batch), B)
        chunks = [(c0, min(c0 + mb, B)) for c0 in range(0, B, mb)]
        # loss normalisation over the WHOLE batch (train.py:440-446): mean over B*Kp_i*D elements per mask, / n_masks;
        # |grad| is carried as {+-s_i} in bf16 and the common factor alpha is applied in fp32 where parameter
        # gradients are written
        numels = [B * mp.shape[1] * D for mp in masks_pred]
        nmin = min(numels)
        alpha = 1.0 / (nmin * n_masks)
        with_reg = self.reg_coeff != 0.0
        pstd = torch.empty((B, D), dtype=torch.float32, device=self.device)
        side = side_stream(self.device)
        hook = self.reducer.layer_done if self.reducer.enabled else None
        for ci, (c0, c1) in enumerate(chunks):
            last = ci == len(chunks) - 1
            beta = 0.0 if ci == 0 else 1.0
            cl = clips[c0:c1]
            me = [m[c0:c1] for m in masks_enc] if len(chunks) > 1 else masks_enc
            mp = [m[c0:c1] for m in masks_pred] if len(chunks) > 1 else masks_pred
            me = [m if m.is_contiguous() else m.contiguous() for m in me]
            mp = [m if m.is_contiguous() else m.contiguous() for m in mp]
            Bc = c1 - c0
            # ---- forward: the EMA target branch is independent of the context branch until the loss, so it runs on
            #      the side stream concurrently (fills the tails of each other's kernels)
            fwd_overlap = side.enabled and _OVERLAP_FWD
            if fwd_overlap:
                side.fork(cl, *mp)
                with torch.cuda.stream(side.stream):
                    h = self.forward_target(cl, mp)
            else:
                h = self.forward_target(cl, mp)
            z, segs, saved_e = encoder_forward(self.ew, cl, me, save=True, ws_tag=self._ws + "enc_save", gates=self._gate("enc"))
            zhat, tsegs, saved_p = predictor_forward(self.pw, z, segs, me, mp, save=True, ws_tag=self._ws + "pred_save",
                                                     gemm_flags=_PRED_GEMM_FLAGS, gates=self._gate("pred"))
            mark('context+predictor forward (main stream)')
            if fwd_overlap:
                side.join()
                for t in h:
                    t.record_stream(torch.cuda.current_stream())
            # ---- loss (train.py:440-459) and its gradient
            dzhat = torch.empty_like(zhat)
            stats = [torch.empty((Bc, D, 2), dtype=torch.float32, device=self.device) if with_reg else None
                     for _ in tsegs]
            for i, t in enumerate(tsegs):
                zi = zhat[t.row0:t.row0 + t.rows]
                ops.latent_loss(zi, h[i], self._stat[0:1], p=self.loss_exp, out_scale=1.0 / (numels[i] * n_masks),
                                accumulate=(i > 0 or ci > 0), dz=dzhat[t.row0:t.row0 + t.rows], gscale=nmin / numels[i])
                ops.token_pstd(zi, pstd[c0:c1], Bc, t.S, D, accumulate=i > 0, stats=stats[i])
            if with_reg:   # gradient of reg_coeff * mean(relu(1 - pstd)), expressed in the 1/alpha units dzhat carries
                coef = self.reg_coeff / (B * D * n_masks) / alpha
                for i, t in enumerate(tsegs):
                    ops.reg_grad(zhat[t.row0:t.row0 + t.rows], pstd[c0:c1], stats[i], dzhat[t.row0:t.row0 + t.rows], Bc,
                                 t.S, D, n_masks, coef)
            mark('target forward joined + loss')
            # ---- backward (predictor first, then encoder layers L-1..0); on the last micro-batch the gradient
            #      buckets go out as layers finish.  A pending update (overlap_update) must be complete first: the backward
            #      overwrites the gradients it reads and uses the W^T shadows it refreshes last.
            if self._gates is not None:
                torch.cuda.current_stream().wait_event(self._gates["done"])
                self._gates = None
            if last:
                self.reducer.begin(side.stream if side.enabled else None)
            lhook = hook if last else None
            if _PRED_DGRAD_FLAGS:
                from ..hip.lib import set_option
                old_dg = set_option("gemm_dgrad_flags", _PRED_DGRAD_FLAGS)
            dz = predictor_backward(dzhat, saved_p, self.pw, segs, alpha, on_layer_done=lhook, beta=beta,
                                    ws_tag=self._ws + "bwd_tmp")
            if _PRED_DGRAD_FLAGS:
                set_option("gemm_dgrad_flags", old_dg)
            mark('predictor backward (main stream)')
            encoder_backward(dz, saved_e, self.ew, segs, alpha, on_layer_done=lhook, beta=beta,
                             ws_tag=self._ws + "bwd_tmp")
            mark('encoder backward (main stream)')
        ops.reg_finish(pstd, n_masks, self._stat[1:2])
        self.reducer.finish()
        # ---- clip / AdamW / EMA / bf16 re-cast (train.py:461-487)
        clipped = bool(clip_now and self.clip_grad is not None)
        self._last_clip = float(self.clip_grad) if clipped else 0.0
        self.optimizer_step(lr, wd, ema, clip_now)
        mark('wgrad stream joined + AdamW/EMA')
        return StepOutput(self._stat.clone(), self.reg_coeff, lr, wd, ema, clipped, 1.0 / self.world_size)

    def _phase_mark(self, name):
        """Phase timestamps on the main stream (tools / bench diagnostics): set `self.phase_events = []` to collect
        (name, event) pairs for the following steps; None (default) costs nothing."""
        ev = getattr(self, 'phase_events', None)
        if ev is not None:
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            ev.append((name, e))

    # ------------------------------------------------------------------------------------------------ update
    @property
    def opt_step(self):
        """Adam step count t (host view of the device counter; synchronises)."""
        return int(self._step_dev.item())

    @opt_step.setter
    def opt_step(self, v):
        self._step_dev.fill_(float(v))

    def set_ln_fold(self, on):
        """Switch the folded-LayerNorm form of the target forward on / off (between steps).  The folded weights are allocated on
        first use and refreshed here; while the switch is on every optimizer step refreshes them after its EMA update."""
        on = bool(on)
        if on:
            self.sync_update()
            if self._target_folds is None:
                self._target_folds = self.tarena.make_folds(len(self.tw.blocks))
            else:
                self.tarena.refresh_folds(0, len(self.tw.blocks))
        self.ln_fold_target = on

    def _make_update_stream(self):
        """A stream for the deferred update that shares a hardware queue with neither the main nor the side stream (a shared
        queue serialises the two: engine/layers.py independent_stream)."""
        from .layers import independent_stream
        with torch.cuda.device(self.device):
            others = [torch.cuda.current_stream(self.device), side_stream(self.device).stream]
            comm = getattr(getattr(self, "reducer", None), "comm_stream", None)
            if comm is not None:   # (created late, after the reducer picked its communication stream)
                others.append(comm)
            make = (lambda: low_priority_stream(self.device)) if _UPD_LOW_PRIO else None
            return independent_stream(self.device, others, make=make)

    def _gate(self, which):
        return None if self._gates is None else self._gates[which]

    def sync_update(self):
        """Make the current stream wait for a pending update (overlap_update): call before reading parameters, moments or
        the EMA target with plain torch operators right after train_step.  No-op otherwise."""
        if self._gates is not None:
            cur = torch.cuda.current_stream()
            cur.wait_event(self._gates["done"])
            # only the stream the step runs on may retire the gates: called under a user's stream (logging, checkpointing), the next
            # train_step on the training stream still has to wait for the update (round-5 advisor finding)
            if self._gates.get("stream") is None or cur == self._gates["stream"]:
                self._gates = None

    def _update_plan(self):
        """[(which, first_block, [(group, lo, hi)])]: the arena ranges of the fused update in the order the forward reads
        them -- encoder blocks in a few ranges of growing size (the first holds the patch embedding, the last the final
        norm; each is one slice of the decayed and one of the no-decay group, which the EMA target arena mirrors), then the
        predictor.  Falls back to the four whole groups if the arena is not laid out in forward order."""
        A = self.arena
        depth = len(self.ew.blocks)
        cuts = sorted({min(depth, max(1, math.ceil(depth * f))) for f in (1.0 / 12, 0.25, 0.5, 1.0)})
        firsts = [0] + cuts[:-1]

        def stage_of(name):   # enc.<...>: which range of the plan the tensor belongs to
            parts = name.split(".")
            if parts[1] == "blocks":
                b = int(parts[2])
                return max(i for i, f in enumerate(firsts) if f <= b)
            return 0 if parts[1] == "patch_embed" else len(firsts) - 1
        plan, ok = [], True
        per_stage = [[] for _ in firsts]
        for gi in (0, 1):
            lo, hi = A.group_ranges[gi]
            slots = sorted((s for n, s in A.slots.items() if n.startswith("enc.") and lo <= s.off < hi), key=lambda s: s.off)
            stages = [stage_of(s.name) for s in slots]
            ok = ok and stages == sorted(stages) and set(stages) == set(range(len(firsts)))
            if not ok:
                break
            starts = [min(s.off for s, st in zip(slots, stages) if st == k) for k in range(len(firsts))]
            starts[0] = lo
            for k in range(len(firsts)):
                per_stage[k].append((gi, starts[k], starts[k + 1] if k + 1 < len(firsts) else hi))
        if ok:
            plan = [("enc", firsts[k], per_stage[k]) for k in range(len(firsts))]
        else:
            plan = [("enc", 0, [(gi,) + tuple(A.group_ranges[gi]) for gi in (0, 1)])]
        plan.append(("pred", 0, [(gi,) + tuple(A.group_ranges[gi]) for gi in (2, 3)]))
        return plan

    def _enc_pred_ranges(self):
        A = self.arena
        (e0, _), (_, e1) = A.group_ranges[0], A.group_ranges[1]
        (p0, _), (_, p1) = A.group_ranges[2], A.group_ranges[3]
        return (e0, e1), (p0, p1)

    def optimizer_step(self, lr, wd, ema, clip_now=False):
        """clip_grad_norm_ x2 (when active) + GradScaler's skip-on-non-finite + AdamW + EMA + bf16 re-casts, all decided
        on the device: two sum-of-squares passes over the gradient arena feed the fused update kernel."""
        A = self.arena
        inv_world = 1.0 / self.world_size
        self.sync_update()   # (an update still pending here can only come from a caller that mixes the two entry points)
        (e0, e1), (p0, p1) = self._enc_pred_ranges()
        ops.sqnorm(A.G[e0:e1], self._stat[4:6])
        ops.sqnorm(A.G[p0:p1], self._stat[6:8])
        gstat = self._stat[4:8]
        ops.step_advance(gstat, self._step_dev)
        clip = float(self.clip_grad) if (clip_now and self.clip_grad is not None) else 0.0
        b1, b2 = self.betas
        T = self.tarena

        def update(gi, lo, hi):
            if hi == lo:
                return
            is_enc = gi < 2
            decay = wd if gi in (0, 2) else 0.0
            tgt = T.P[lo - T.lo:hi - T.lo] if is_enc else None
            tgtb = T.Pb[lo - T.lo:hi - T.lo] if is_enc else None
            ops.adamw_ema_guarded(A.P[lo:hi], A.G[lo:hi], A.M1[lo:hi], A.M2[lo:hi], A.Pb[lo:hi], tgt, tgtb, lr, decay, b1,
                                  b2, self.eps, inv_world, ema, gstat, 0 if is_enc else 1, clip, inv_world,
                                  self._step_dev)

        if self.overlap_update and side_stream(self.device).enabled:
            # range by range on the update stream, in the order the next forward reads the weights; one event per range
            main = torch.cuda.current_stream()
            ready = torch.cuda.Event()
            ready.record(main)          # gradients, norms and the step count are final
            if self._upd_stream is None:   # (normally created by the constructor; here after the flag was flipped between steps)
                self._upd_stream = self._make_update_stream()
            upd = self._upd_stream
            upd.wait_event(ready)
            gates = {"enc": [], "pred": []}
            with torch.cuda.stream(upd):
                n_enc = sum(1 for w, _, _ in self._plan if w == "enc")
                firsts = [f for w, f, _ in self._plan if w == "enc"] + [len(self.tw.blocks)]
                for pi, (which, first_block, ranges) in enumerate(self._plan):
                    for gi, lo, hi in ranges:
                        update(gi, lo, hi)
                    if which == "enc" and self.ln_fold_target:   # the folded target weights of this range's blocks, before its gate
                        T.refresh_folds(first_block, firsts[pi + 1] if pi + 1 <= n_enc else len(self.tw.blocks))
                    ev = torch.cuda.Event()
                    ev.record(upd)
                    gates[which].append((first_block, ev))
                A.refresh_transposed()
                done = torch.cuda.Event()
                done.record(upd)
            gates["done"] = done
            gates["stream"] = main
            self._gates = gates
            _weights.PENDING_UPDATE[self.device.index] = done
        else:
            for gi, (lo, hi) in enumerate(A.group_ranges):
                update(gi, lo, hi)
            if self.ln_fold_target:
                T.refresh_folds(0, len(self.tw.blocks))
            A.refresh_transposed()
        bump_generation()   # parameters changed behind torch's version counters: derived-weight caches must refresh
        for g in self.param_groups:
            g["lr"] = lr
            if not g.get("WD_exclude", False):
                g["weight_decay"] = wd

    # ------------------------------------------------------------------------------------------------ logging
    def arena_stats(self):
        """Per-tensor gradient norms and Adam-moment magnitudes of the step that just ran: one launch over the arenas
        (vj_grad_stats_multi) + one copy.  {'enc'|'pred': {'grads': [(name, norm, is_matrix)], 'moments': [(mean|m|,
        mean|v|)]}}; gradients are the data-parallel averages (what DDP leaves in p.grad, train.py:476-479).
        Cached until the next train_step."""
        if self._arena_stats is not None:
            return self._arena_stats
        self.sync_update()
        if self._gs_desc is None:
            order = [("enc", a, n, p) for a, n, p in self._enc_named if p.requires_grad]
            order += [("pred", a, n, p) for a, n, p in self._pred_named if p.requires_grad]
            # "weight matrix" exactly as the reference's grad_logger filters (src/utils/logging.py:91-105): not a
            # `.bias`, not 1-D -- which is NOT the optimizer's no-decay rule ('bias' anywhere in the name)
            self._gs_meta = [(which, n, self.arena.slots[a].numel, not (n.endswith('.bias') or p.dim() == 1))
                             for which, a, n, p in order]
            flat = []
            for which, a, n, p in order:
                flat += [self.arena.slots[a].off, self.arena.slots[a].numel]
            self._gs_desc = torch.tensor(flat, dtype=torch.int64, device=self.device)
        A = self.arena
        sums = ops.grad_stats_multi(A.G, A.M1, A.M2, self._gs_desc, len(self._gs_meta)).tolist()
        inv_world = 1.0 / self.world_size
        # the reference logs AFTER clip_grad_norm_ (train.py:466-481): the arena keeps the unclipped gradients (the clip
        # coefficient is applied inside the fused AdamW kernel), so the same coefficient is applied to the logged norms
        fac = {"enc": 1.0, "pred": 1.0}
        clip = getattr(self, "_last_clip", 0.0)
        if clip > 0.0:
            h = self._stat.tolist()
            for which, i in (("enc", 4), ("pred", 6)):
                fac[which] = min(1.0, clip / (math.sqrt(h[i]) * inv_world + 1e-6))
        out = {"enc": {"grads": [], "moments": []}, "pred": {"grads": [], "moments": []}}
        for (which, n, numel, is_mat), (sq, a1, a2) in zip(self._gs_meta, sums):
            out[which]["grads"].append((n, math.sqrt(sq) * inv_world * fac[which], is_mat))
            out[which]["moments"].append((a1 / numel, a2 / numel))
        self._arena_stats = out
        return out

    def zero_grad(self, set_to_none=False):
        """Gradients are fully overwritten by every backward (beta = 0 wgrads); kept for API compatibility."""
        return None

    # ------------------------------------------------------------------------------------------------ checkpoints
    def state_dict(self):
        """torch.optim.AdamW-compatible optimizer state (reference checkpoint key 'opt', train.py:331-342): parameter ids
        count every member of the reference's four groups (frozen position tables included, stateless)."""
        self.sync_update()
        return optstate.build_state_dict(self.param_groups, self._slot_of, self.arena.M1, self.arena.M2, self.opt_step,
                                         self.betas, self.eps)

    def load_state_dict(self, sd):
        """Accepts our own state_dict() and the 'opt' entry of a reference checkpoint (same grouping and ids)."""
        self.sync_update()
        step = optstate.load_state_dict(self.param_groups, self._slot_of, self.arena.M1, self.arena.M2, sd)
        if step is not None:
            self.opt_step = step

    def sync_shadows(self):
        """Call after writing parameters from outside (load_state_dict): refresh bf16 / transposed shadows."""
        self.sync_update()
        self.arena.refresh_bf16()
        self.arena.refresh_transposed()
        self.tarena.refresh_bf16()
        self.tarena.refresh_folds(0, len(self.tw.blocks))
        bump_generation()
