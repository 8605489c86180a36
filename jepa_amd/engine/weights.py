"""Weight storage for the HIP path: flat fp32 arenas (master weights, gradients, Adam moments, EMA target) with
bf16 shadows (the MFMA GEMM operands) and transposed bf16 shadows (dgrad operands), plus the per-layer view
structs the forward/backward chains consume.

Layout (one contiguous fp32 arena, every tensor padded to 64 elements so all bf16 views are 16-byte aligned):

    [ encoder decayed | encoder bias/1-D | predictor decayed | predictor bias/1-D ]

which makes each of the four AdamW parameter groups of the reference (app/vjepa/utils.py:173-191) one contiguous
range -> one fused kernel launch per group, and the encoder range lines up 1:1 with the EMA target arena
(train.py:483-487).  Gradients are written by the wgrad GEMMs / reduction kernels straight into the gradient
arena; `param.grad` of every module parameter is a view into it, `param.data` a view into the master arena.
"""
from dataclasses import dataclass, field
from typing import List, Optional

import torch

from ..hip import ops

# Bumped whenever parameters are rewritten through raw arena pointers (the fused AdamW / EMA kernels, shadow refreshes):
# torch's per-tensor `_version` counters do not see those writes, so every cache of derived weights (bf16 copies of
# stand-alone modules, private arenas of engine/hipmodule.py) keys on this generation as well.
ARENA_GENERATION = [0]


def bump_generation():
    ARENA_GENERATION[0] += 1


# Trainer(overlap_update=True) issues the fused AdamW / EMA update on its own stream and returns before it has run; the NEXT
# train_step orders itself against it range by range.  Every OTHER reader of the parameters inside this package (module-level
# forwards, logging, checkpoints) calls wait_pending_update() first: the current stream then waits for the last update.
# Code outside the package that reads `param.data` with torch operators must call `trainer.sync_update()` itself.
PENDING_UPDATE = {}   # device index -> torch.cuda.Event recorded after the last range (and the W^T refresh) of the update


def wait_pending_update(device=None):
    if not PENDING_UPDATE:
        return
    idx = torch.cuda.current_device() if device is None or torch.device(device).index is None else torch.device(device).index
    ev = PENDING_UPDATE.get(idx)
    if ev is not None:
        torch.cuda.current_stream(idx).wait_event(ev)


from .optstate import ALIGN, Slot as _Slot, is_no_decay, layout, pad64 as _pad  # noqa: F401  (pure, CPU-testable)


# ----------------------------------------------------------------------------------------------- view structs
@dataclass
class LinearW:
    w: torch.Tensor                       # bf16 [N, K]
    b: Optional[torch.Tensor]             # fp32 [N]
    wT: Optional[torch.Tensor] = None     # bf16 [K, N]  (dgrad operand; None for inference-only weights)
    gw: Optional[torch.Tensor] = None     # fp32 [N, K]  gradient view
    gb: Optional[torch.Tensor] = None     # fp32 [N]


@dataclass
class NormW:
    g: torch.Tensor
    b: torch.Tensor
    gg: Optional[torch.Tensor] = None
    gb: Optional[torch.Tensor] = None


@dataclass
class BlockW:
    norm1: NormW
    qkv: LinearW
    proj: LinearW
    norm2: NormW
    fc1: LinearW
    fc2: LinearW


@dataclass
class FoldW:
    """One block's LayerNorms folded into the Linears that consume them (vj_lnfold_t): for trunks that never run backward."""
    w_qkv: torch.Tensor                   # bf16 [3D, D] = bf16(W_qkv * norm1.weight)
    c_qkv: torch.Tensor                   # fp32 [3D]  row sums of w_qkv
    b_qkv: torch.Tensor                   # fp32 [3D]  qkv.bias + W_qkv norm1.bias
    w_fc1: torch.Tensor                   # bf16 [Dh, D]
    c_fc1: torch.Tensor
    b_fc1: torch.Tensor


@dataclass
class EncoderW:
    patch: LinearW                        # Conv3d weight viewed [D, C*tub*p*p]
    pos: torch.Tensor                     # fp32 [N, D]
    blocks: List[BlockW]
    norm: NormW
    heads: int = 1
    tubelet: int = 2
    patch_size: int = 16
    folds: Optional[List[FoldW]] = None   # set (EMA target encoder, option ln_fold): the forward takes vj_blocks_fwd_lnfold


@dataclass
class PredictorW:
    embed: LinearW
    mask_tokens: List[torch.Tensor]       # fp32 [Dp] each
    g_mask_tokens: List[Optional[torch.Tensor]]
    pos: torch.Tensor                     # fp32 [N, Dp]
    blocks: List[BlockW]
    norm: NormW
    proj: LinearW
    heads: int = 1


# ----------------------------------------------------------------------------------------------- arena
class ParamArena:
    """Flat fp32 master/grad/moment arenas + bf16 shadows for a list of (name, Parameter) groups."""

    def __init__(self, groups, device, with_moments=True, bind_grads=True):
        """groups: list of lists of (name, param); each group becomes one contiguous, 64-padded range."""
        self.device = device
        self.slots, self.group_ranges, off = layout(groups)
        self.total = off
        self.P = torch.zeros(off, dtype=torch.float32, device=device)
        self.G = torch.zeros(off, dtype=torch.float32, device=device)
        self.Pb = torch.zeros(off, dtype=torch.bfloat16, device=device)
        self.M1 = torch.zeros(off, dtype=torch.float32, device=device) if with_moments else None
        self.M2 = torch.zeros(off, dtype=torch.float32, device=device) if with_moments else None
        self.wT = {}
        self.frozen = {}  # name -> fp32 device tensor (frozen tables such as pos_embed)
        with torch.no_grad():
            for s in self.slots.values():
                self.P[s.off:s.off + s.numel].copy_(s.param.data.reshape(-1).to(device=device, dtype=torch.float32))
                s.param.data = self.P[s.off:s.off + s.numel].view(s.shape)
                if bind_grads:
                    s.param.grad = self.G[s.off:s.off + s.numel].view(s.shape)
        self.refresh_bf16()

    # -- views
    def f32(self, name):
        s = self.slots[name]
        return self.P[s.off:s.off + s.numel].view(s.shape)

    def grad(self, name):
        s = self.slots[name]
        return self.G[s.off:s.off + s.numel].view(s.shape)

    def bf16(self, name):
        s = self.slots[name]
        return self.Pb[s.off:s.off + s.numel].view(s.shape)

    def refresh_bf16(self):
        ops.cast_bf16(self.P, self.Pb)

    def make_transposed(self, names):
        """Allocate + fill transposed bf16 shadows W^T for the given 2-D weights (views [N,K] -> [K,N])."""
        for n in names:
            w = self.bf16(n)
            w2 = w.reshape(w.shape[0], -1)
            self.wT[n] = torch.empty((w2.shape[1], ops.pad64(w2.shape[0])), dtype=torch.bfloat16, device=self.device)
        self.refresh_transposed()

    def _build_transpose_plan(self):
        """Device-side descriptor / block tables for the single-launch refresh of every W^T shadow."""
        desc, blocks = [], []
        for ti, (n, t) in enumerate(self.wT.items()):
            w = self.bf16(n)
            w2 = w.reshape(w.shape[0], -1)
            M, N = w2.shape                      # W [N_out, K_in] -> W^T [K_in, pad64(N_out)]
            Mp = t.shape[1]
            desc += [w2.data_ptr(), t.data_ptr(), M, N, w2.stride(0), Mp]
            for tm in range((Mp + 63) // 64):
                for tn in range((N + 63) // 64):
                    blocks += [ti, tm, tn, 0]
        self._tp_desc = torch.tensor(desc, dtype=torch.int64, device=self.device)
        self._tp_blocks = torch.tensor(blocks, dtype=torch.int32, device=self.device)
        self._tp_n = len(blocks) // 4
        self._tp_key = tuple(self.wT.keys())

    def refresh_transposed(self):
        if not self.wT:
            return
        if getattr(self, "_tp_key", None) != tuple(self.wT.keys()):
            self._build_transpose_plan()
        ops.transpose_multi(self._tp_desc, self._tp_blocks, self._tp_n)


def _lin(arena, prefix, train, w_shape2d=None):
    w = arena.bf16(prefix + ".weight")
    if w_shape2d is not None:
        w = w.reshape(w_shape2d)
    b = arena.f32(prefix + ".bias") if (prefix + ".bias") in arena.slots else None
    if not train:
        return LinearW(w=w, b=b)
    wT = arena.wT.get(prefix + ".weight")
    if wT is not None:
        wT = wT[:, :w.shape[0]]  # [K_in, N_out] view of the 64-padded transposed shadow
    gw = arena.grad(prefix + ".weight")
    if w_shape2d is not None:
        gw = gw.reshape(w_shape2d)
    return LinearW(w=w, b=b, wT=wT, gw=gw, gb=arena.grad(prefix + ".bias") if b is not None else None)


def _norm(arena, prefix, train):
    if not train:
        return NormW(arena.f32(prefix + ".weight"), arena.f32(prefix + ".bias"))
    return NormW(arena.f32(prefix + ".weight"), arena.f32(prefix + ".bias"), arena.grad(prefix + ".weight"),
                 arena.grad(prefix + ".bias"))


def _block(arena, prefix, train):
    return BlockW(norm1=_norm(arena, prefix + "norm1", train), qkv=_lin(arena, prefix + "attn.qkv", train),
                  proj=_lin(arena, prefix + "attn.proj", train), norm2=_norm(arena, prefix + "norm2", train),
                  fc1=_lin(arena, prefix + "mlp.fc1", train), fc2=_lin(arena, prefix + "mlp.fc2", train))


def linear_weight_names(module_prefix, n_blocks, block_prefix, extra=()):
    names = []
    for i in range(n_blocks):
        for l in ("attn.qkv", "attn.proj", "mlp.fc1", "mlp.fc2"):
            names.append(f"{module_prefix}{block_prefix}.{i}.{l}.weight")
    names += [module_prefix + e for e in extra]
    return names


def encoder_views(arena, prefix, vit, pos, train):
    """prefix: name prefix of the VisionTransformer's parameters inside the arena (e.g. 'enc.')."""
    D = vit.embed_dim
    return EncoderW(patch=_lin(arena, prefix + "patch_embed.proj", train, (D, -1)), pos=pos,
                    blocks=[_block(arena, f"{prefix}blocks.{i}.", train) for i in range(len(vit.blocks))],
                    norm=_norm(arena, prefix + "norm", train), heads=vit.num_heads, tubelet=vit.tubelet_size,
                    patch_size=vit.patch_size)


def predictor_views(arena, prefix, pred, pos, train):
    n_tok = pred.num_mask_tokens
    toks = [arena.f32(f"{prefix}mask_tokens.{i}").reshape(-1) for i in range(n_tok)]
    gtoks = [arena.grad(f"{prefix}mask_tokens.{i}").reshape(-1) if train else None for i in range(n_tok)]
    return PredictorW(embed=_lin(arena, prefix + "predictor_embed", train), mask_tokens=toks, g_mask_tokens=gtoks,
                      pos=pos,
                      blocks=[_block(arena, f"{prefix}predictor_blocks.{i}.", train)
                              for i in range(len(pred.predictor_blocks))],
                      norm=_norm(arena, prefix + "predictor_norm", train), proj=_lin(arena, prefix + "predictor_proj", train),
                      heads=pred.num_heads)
