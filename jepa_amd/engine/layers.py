"""Hand-written forward / backward chains of the V-JEPA encoder and predictor over the C-ABI kernels.

No autograd graph: every chain saves exactly the activations its backward needs and the backward walks the layers
in reverse, writing weight gradients (fp32) straight into the gradient-arena views.  Both masks of a step run
through ONE chain: their token rows are concatenated along M (bigger GEMMs, each weight gradient produced in one
piece -- which is also what makes a layer's gradient bucket final as soon as that layer's backward is done), and
only attention, which is per-sequence, is launched per mask segment.

Reference semantics restated (never copied):
  Block / Attention / MLP        src/models/utils/modules.py:13-120
  VisionTransformer.forward      src/models/vision_transformer.py:159-195
  VisionTransformerPredictor     src/models/predictor.py:174-239
  MultiMask wrappers             src/models/utils/multimask.py:11-48
"""
import os
from dataclasses import dataclass
from typing import List, Optional

import torch

from ..hip import ops
from . import chain
from .weights import BlockW, EncoderW, LinearW, PredictorW

LN_EPS = 1e-6  # partial(nn.LayerNorm, eps=1e-6), vision_transformer.py:252-281 / predictor.py:242-246


class SideStream:
    """Second HIP stream of the step: the EMA-target forward in the forward phase, the weight-gradient GEMMs (and the
    reductions feeding the bias gradients) in the backward phase -- work that is off the main chain and fills the CUs its
    kernels leave idle.  `fork` makes the side stream wait for everything enqueued so far on the main stream; `join` makes
    the main stream wait for the side stream.  (A separate lowest-priority stream for the weight gradients was measured in
    round 3: no effect, +0.1 % in 7 of 8 interleaved rounds -- both phases are throughput-bound, not dispatch-order-bound.)"""

    def __init__(self, device):
        import os
        with torch.cuda.device(device):
            self.stream = independent_stream(device, [torch.cuda.current_stream(device)])
        self.enabled = os.environ.get("VJ_NO_OVERLAP", "0") != "1"   # serial mode for per-kernel profiling

    def fork(self, *tensors):
        self.stream.wait_stream(torch.cuda.current_stream())
        for t in tensors:          # allocator plumbing: these buffers are read on the side stream
            if t is not None:
                t.record_stream(self.stream)

    def join(self):
        torch.cuda.current_stream().wait_stream(self.stream)


def low_priority_stream(device):
    """A HIP stream of the LOWEST priority the device offers (torch.cuda.Stream only reaches the default and higher ones), wrapped
    for torch: the workgroups of the deferred fused update should take CUs only where the two forward streams leave them.
    Falls back to a plain stream when the runtime call is unavailable."""
    import ctypes
    dev = torch.device(device)
    try:
        hip = ctypes.CDLL("libamdhip64.so")
        least, greatest = ctypes.c_int(0), ctypes.c_int(0)
        with torch.cuda.device(dev):
            if hip.hipDeviceGetStreamPriorityRange(ctypes.byref(least), ctypes.byref(greatest)) != 0:
                raise OSError("hipDeviceGetStreamPriorityRange")
            h = ctypes.c_void_p()
            if hip.hipStreamCreateWithPriority(ctypes.byref(h), 1, least.value) != 0 or not h.value:   # 1 = hipStreamNonBlocking
                raise OSError("hipStreamCreateWithPriority")
        return torch.cuda.ExternalStream(h.value, device=dev)
    except (OSError, AttributeError):
        return torch.cuda.Stream(device=dev)


_INDEP_LOG = []   # (candidate index, [concurrent with others[i]]) of every independent_stream() call: diagnostics / tests


def streams_concurrent(a, b, spin_us=300.0):
    """True when a kernel on stream `a` and one on stream `b` run at the same time (different hardware queues): two idle one-wave
    kernels of `spin_us` each (vj_probe_spin_stamped) leave their start / end stamps of the chip-wide 100 MHz timer in device memory;
    the streams are concurrent when the two intervals overlap by more than half a spin.  Judged on the DEVICE's clock (round 6; it was
    host wall-clock time around both, which eight ranks and their loader workers can stretch past any threshold).  Synchronises the
    two streams."""
    from ..hip.lib import check, load_library
    lib = load_library()
    ticks = int(spin_us * 100)
    dev = a.device if hasattr(a, "device") else torch.device("cuda", torch.cuda.current_device())
    stamps = torch.zeros(4, dtype=torch.int64, device=dev)
    torch.cuda.current_stream(dev).synchronize()      # the zero fill is complete before either probe writes
    best = 0
    for _ in range(2):                       # the first pass also pays one-time costs (code load, queue creation)
        check(lib.vj_probe_spin_stamped(ticks, stamps.data_ptr(), a.cuda_stream), "vj_probe_spin_stamped")
        check(lib.vj_probe_spin_stamped(ticks, stamps.data_ptr() + 16, b.cuda_stream), "vj_probe_spin_stamped")
        a.synchronize()
        b.synchronize()
        a0, a1, b0, b1 = stamps.tolist()
        best = max(best, min(a1, b1) - max(a0, b0))
    return best > ticks // 2


def independent_stream(device, others, make=None, tries=16):
    """A stream that shares a hardware queue with none of `others` (torch streams).  ROCclr maps streams onto GPU_MAX_HW_QUEUES
    hardware queues round-robin, and torch hands out its 32 pooled streams round-robin, so the n-th stream a process creates may
    land on the queue of the main or the side stream -- kernels of the two then serialise (measured: 72 -> 86 ms per step when the
    deferred update's stream aliased a forward stream, round 5; the same signature as the unexplained vj_comm_* slowdown of round 4).
    Candidates come from `make()` (default: torch.cuda.Stream) and are tested with streams_concurrent; the last one is returned
    with a warning if none is independent (GPU_MAX_HW_QUEUES too small)."""
    dev = torch.device(device)
    make = make or (lambda: torch.cuda.Stream(device=dev))
    cand = None
    for i in range(tries):
        cand = make()
        ok = [streams_concurrent(cand, o) for o in others]
        _INDEP_LOG.append((i, ok))
        if all(ok):
            return cand
    import warnings
    warnings.warn("jepa_amd: no stream independent of the existing ones was found (GPU_MAX_HW_QUEUES too small?); "
                  "streams will serialise on a shared hardware queue")
    return cand


_SIDE = {}


def side_stream(device):
    """One SideStream per GPU, keyed by device INDEX: torch.device('cuda') and torch.device('cuda:0') are different
    dictionary keys but the same GPU, and the fork/join partners must be the same stream object."""
    device = torch.device(device)
    idx = device.index if device.index is not None else torch.cuda.current_device()
    s = _SIDE.get(idx)
    if s is None:
        s = _SIDE[idx] = SideStream(torch.device("cuda", idx))
    return s


# The C launch chains (vj_blocks_fwd / vj_blocks_bwd) are the default for the Trainer; VJ_PY_CHAIN=1 keeps the
# per-kernel Python chain below (same kernels, same order: bit-identical results, ~10x the host time).
USE_C_CHAIN = os.environ.get("VJ_PY_CHAIN", "0") != "1"


@dataclass
class Seg:
    """A group of B equal-length sequences occupying rows [row0, row0 + B*S) of the token matrix."""
    row0: int
    B: int
    S: int

    @property
    def rows(self):
        return self.B * self.S


def _rows(t, seg):
    return t[seg.row0:seg.row0 + seg.rows]


def _q_prescaled(D: int) -> bool:
    from ..hip.lib import get_option
    return get_option("attn_softmax") == 2 and D % 4 == 0


def _f32_mul(a: float, b: float) -> float:
    """a * b rounded as the C chain rounds it (float * float)."""
    import numpy as np
    return float(np.float32(np.float32(a) * np.float32(b)))


# =============================================================================================== block
def block_forward(x, bw: BlockW, segs: List[Seg], heads: int, save: bool, fold=None):
    """x [M, D] bf16 -> x2 [M, D]; returns (x2, saved).  fold (FoldW, only with save = False): both LayerNorms folded into the
    GEMMs that consume them -- the kernels and their order are those of vj_blocks_fwd_lnfold (bit-identical results)."""
    D = x.shape[1]
    hd = D // heads
    scale = hd ** -0.5
    if fold is not None and save:
        raise ValueError("block_forward: folded LayerNorms keep nothing for a backward")
    y1 = mean1 = rstd1 = None
    if fold is not None:
        rs = ops.ln_rowstats(x, LN_EPS)
        if _q_prescaled(D):
            qkv = ops.gemm_nt_lnfold(x, fold.w_qkv, fold.b_qkv, rs, fold.c_qkv, epilogue=ops.EPI_QKV,
                                     alpha=_f32_mul(scale, 1.4426950408889634))
            scale = -scale
        else:
            qkv = ops.gemm_nt_lnfold(x, fold.w_qkv, fold.b_qkv, rs, fold.c_qkv)
    elif _q_prescaled(D):   # option attn_softmax = 2 (as vj_blocks_fwd): the q third of qkv carries scale * log2(e)
        y1, mean1, rstd1 = ops.layernorm_fwd(x, bw.norm1.g, bw.norm1.b, LN_EPS, save_stats=save)
        qkv = ops.gemm_nt(y1, bw.qkv.w, bias=bw.qkv.b, epilogue=ops.EPI_QKV, alpha=_f32_mul(scale, 1.4426950408889634))
        scale = -scale    # "q is pre-scaled" for the attention entry points
    else:
        y1, mean1, rstd1 = ops.layernorm_fwd(x, bw.norm1.g, bw.norm1.b, LN_EPS, save_stats=save)
        qkv = ops.gemm_nt(y1, bw.qkv.w, bias=bw.qkv.b)
    o = torch.empty_like(x)
    lses = []
    for sg in segs:
        _, lse = ops.attn_fwd(_rows(qkv, sg), sg.B, sg.S, heads, hd, scale, save_lse=save, out=_rows(o, sg))
        lses.append(lse)
    x1 = ops.gemm_nt(o, bw.proj.w, bias=bw.proj.b, residual=x)
    y2 = mean2 = rstd2 = u = None
    if fold is not None:
        g = ops.gemm_nt_lnfold(x1, fold.w_fc1, fold.b_fc1, ops.ln_rowstats(x1, LN_EPS), fold.c_fc1, epilogue=ops.EPI_GELU)
    else:
        y2, mean2, rstd2 = ops.layernorm_fwd(x1, bw.norm2.g, bw.norm2.b, LN_EPS, save_stats=save)
        # u: the fc1 epilogue saves gelu'(pre-activation) here (not the pre-activation): all the backward needs from it
        u = torch.empty((x.shape[0], bw.fc1.w.shape[0]), dtype=torch.bfloat16, device=x.device) if save else None
        g = ops.gemm_nt(y2, bw.fc1.w, bias=bw.fc1.b, aux_out=u, epilogue=ops.EPI_GELU)
    x2 = ops.gemm_nt(g, bw.fc2.w, bias=bw.fc2.b, residual=x1)
    saved = (x, y1, mean1, rstd1, qkv, o, lses, x1, y2, mean2, rstd2, u, g) if save else None
    return x2, saved


# Weight gradients: transpose-free by default (vj_gemm_bf16_tn_splitk reads dY and X token-major as the backward produced
# them; round-3 interleaved A/B at ViT-L B=24: 86.38 vs 86.83 ms/step, faster in 8 of 8 rounds, -25 W, and the 294
# transpose launches / 35 GB per step of the NT route are gone -- profiles/r03_abab_switches.md).  The run-time option
# "wgrad_tn" = 0 (VJ_WGRAD_TN=0) selects the transposes + NT split-K route, kept as the cross-check of the TN kernel.
from ..hip.lib import get_option as _get_option  # noqa: E402


def _tn_ok(n_out: int, k_in: int) -> bool:
    return _get_option("wgrad_tn") != 0 and n_out % 8 == 0 and k_in % 8 == 0   # as chain.hip: any non-zero value is "on"


def _wgrad(dy, x_in, lw: LinearW, alpha: float, beta: float = 0.0, bias_done: bool = False):
    """bias_done: the bias gradient (column sums of dy) was produced by the LayerNorm backward that produced dy."""
    acc = beta != 0.0
    if _tn_ok(dy.shape[1], x_in.shape[1]):
        if lw.gb is not None and not bias_done:
            ops.colsum(dy, lw.gb, alpha=alpha, accumulate=acc)
        ops.gemm_wgrad_tn(dy, x_in, lw.gw, alpha=alpha, beta=beta)
        return
    # (bias_done: the LayerNorm backward already wrote lw.gb -- no second column sum, exactly as chain.hip wgrad() nulls gb)
    need_cs = lw.gb is not None and not bias_done
    dyT = ops.transpose_colsum(dy, lw.gb, alpha=alpha, accumulate=acc) if need_cs else ops.transpose(dy)
    xT = ops.transpose(x_in)
    ops.gemm_wgrad(dyT, xT, lw.gw, alpha=alpha, beta=beta)


def _group_ok(bw) -> bool:
    """Option wgrad_group: the four weight gradients of a block as one vj_gemm_bf16_tn_grouped launch (as vj_blocks_bwd)."""
    return (_get_option("wgrad_tn") != 0 and _get_option("wgrad_group") != 0 and
            all(d % 8 == 0 for lw in (bw.qkv, bw.proj, bw.fc1, bw.fc2) for d in lw.gw.shape))


def _wgrad_group(items, alpha: float, beta: float):
    """items: [(dy, x_in, lw, bias_done)] in the order fc2, fc1, proj, qkv: bias column sums first, then one launch."""
    def run():
        for dy, x_in, lw, bias_done in items:
            if lw.gb is not None and not bias_done:
                ops.colsum(dy, lw.gb, alpha=alpha, accumulate=beta != 0.0)
        ops.gemm_wgrad_tn_grouped([(dy, x_in, lw.gw) for dy, x_in, lw, _ in items], alpha=alpha, beta=beta)
    side = side_stream(items[0][0].device)
    if side.enabled:
        side.fork(*[t for dy, x_in, _, _ in items for t in (dy, x_in)])
        with torch.cuda.stream(side.stream):
            run()
    else:
        run()


def _linear_backward(dy, x_in, lw: LinearW, alpha: float, need_dx=True, dgelu_aux=None, beta: float = 0.0,
                     bias_done: bool = False, defer=None):
    """dW (fp32, into lw.gw) = alpha * dy^T x_in + beta * dW ; db likewise ; returns dx = dy W (bf16).
    The weight-gradient half goes to the side stream; the caller joins before the gradients are consumed.
    defer (list): only record the weight-gradient problem; the caller launches the block's group (_wgrad_group)."""
    side = side_stream(dy.device)
    if defer is not None:
        defer.append((dy, x_in, lw, bias_done))
    elif side.enabled:
        side.fork(dy, x_in)
        with torch.cuda.stream(side.stream):
            _wgrad(dy, x_in, lw, alpha, beta, bias_done)
    else:
        _wgrad(dy, x_in, lw, alpha, beta, bias_done)
    if not need_dx:
        return None
    if dgelu_aux is not None:
        return ops.gemm_nt(dy, lw.wT, aux_in=dgelu_aux, epilogue=ops.EPI_DGELU)
    return ops.gemm_nt(dy, lw.wT)


def block_backward(dx2, saved, bw: BlockW, segs: List[Seg], heads: int, alpha: float, beta: float = 0.0,
                   fc2_bias_done: bool = False, prev_fc2_gb=None):
    """With transpose-free weight gradients the bias gradients of proj and of the PREVIOUS block's fc2 are the column
    sums of the two LayerNorm backward outputs and come out of those passes (vj_layernorm_bwd_colsum); `fc2_bias_done`
    says that this block's fc2 bias gradient was produced that way by the block above.  Same kernels, same order as
    vj_blocks_bwd (csrc/chain.hip): bit-identical results."""
    x, y1, mean1, rstd1, qkv, o, lses, x1, y2, mean2, rstd2, u, g = saved
    D = x.shape[1]
    hd = D // heads
    scale = hd ** -0.5
    if _q_prescaled(D):
        scale = -scale       # the forward stored q pre-scaled (option attn_softmax = 2)
    acc = beta != 0.0
    fuse = _tn_ok(8, 8)
    defer = [] if _group_ok(bw) else None
    du = _linear_backward(dx2, g, bw.fc2, alpha, dgelu_aux=u, beta=beta, bias_done=fuse and fc2_bias_done, defer=defer)   # fc2 dgrad fused with GELU'
    dy2 = _linear_backward(du, y2, bw.fc1, alpha, beta=beta, defer=defer)
    dx1 = ops.layernorm_bwd(dy2, x1, bw.norm2.g, mean2, rstd2, bw.norm2.gg, bw.norm2.gb, dres=dx2, alpha=alpha,
                            accumulate=acc, dxsum=bw.proj.gb if fuse else None)
    do = _linear_backward(dx1, o, bw.proj, alpha, beta=beta, bias_done=fuse, defer=defer)
    dqkv = torch.empty_like(qkv)
    for sg, lse in zip(segs, lses):
        ops.attn_bwd(_rows(qkv, sg), _rows(o, sg), _rows(do, sg), lse, sg.B, sg.S, heads, hd, scale,
                     out=_rows(dqkv, sg))
    if defer is not None:
        defer.append((dqkv, y1, bw.qkv, False))
        _wgrad_group(defer, alpha, beta)
        dy1 = ops.gemm_nt(dqkv, bw.qkv.wT)
    else:
        dy1 = _linear_backward(dqkv, y1, bw.qkv, alpha, beta=beta)
    return ops.layernorm_bwd(dy1, x, bw.norm1.g, mean1, rstd1, bw.norm1.gg, bw.norm1.gb, dres=dx1, alpha=alpha,
                             accumulate=acc, dxsum=prev_fc2_gb if fuse else None)


# =============================================================================================== encoder
def _wait_gates(gates, chained):
    """gates [(first_block, event)]: the weights of blocks >= first_block are final after the event (the previous step's
    range-by-range update, engine/step.py).  The entry gate (block 0: embedding + first range) is waited for here; with the
    C chain the later ones are waited for between the ranges of the trunk (chain.blocks_forward), otherwise all of them now."""
    if not gates:
        return
    cur = torch.cuda.current_stream()
    for b, ev in gates:
        if b == 0 or not chained:
            cur.wait_event(ev)


def encoder_forward(ew: EncoderW, clips, masks: Optional[List[torch.Tensor]], save: bool, final_norm=True, ws_tag=None,
                    gemm_flags=0, gates=None):
    """clips fp32 [B,3,T,H,W]; masks: None (all N tokens) or a list of int64 [B,K_i] index tensors.
    Returns (out [sum_i B*K_i, D] bf16, segs, saved).  With final_norm=False the last residual stream is returned
    (the target path fuses the final norm into vj_target_rows).  gates: see _wait_gates."""
    B = clips.shape[0]
    D = ew.patch.w.shape[0]
    kdim = ew.patch.w.shape[1]
    N = ew.pos.shape[0]
    if masks is None:
        segs = [Seg(0, B, N)]
        tok = ops.tubelet_pack(clips, ew.tubelet, ew.patch_size)
    else:
        segs, r = [], 0
        for m in masks:
            segs.append(Seg(r, B, m.shape[1]))
            r += B * m.shape[1]
        tok = torch.empty((r, kdim), dtype=torch.bfloat16, device=clips.device)
        for sg, m in zip(segs, masks):
            ops.tubelet_pack(clips, ew.tubelet, ew.patch_size, idx=m, out=_rows(tok, sg))
    chained = ws_tag is not None and USE_C_CHAIN
    _wait_gates(gates, chained)
    x = ops.gemm_nt(tok, ew.patch.w, bias=ew.patch.b, flags=(gemm_flags & 0xffff if not gemm_flags >> 16 else 0) or None)
    for i, sg in enumerate(segs):
        ops.add_pos(_rows(x, sg), ew.pos, sg.B, sg.S, idx=None if masks is None else masks[i])
    if chained:    # ws_tag: the caller owns one workspace per trunk (engine/chain.py)
        x, saved_blocks = chain.blocks_forward(x, ew, segs, save, ws_tag, LN_EPS, gemm_flags=gemm_flags, gates=gates)
    else:
        saved_blocks = []
        folds = ew.folds if (ew.folds is not None and not save) else None
        for bi, bw in enumerate(ew.blocks):
            x, sv = block_forward(x, bw, segs, ew.heads, save, fold=None if folds is None else folds[bi])
            saved_blocks.append(sv)
    if not final_norm:
        return x, segs, None
    out, mean, rstd = ops.layernorm_fwd(x, ew.norm.g, ew.norm.b, LN_EPS, save_stats=save)
    saved = (tok, saved_blocks, x, mean, rstd) if save else None
    return out, segs, saved


def encoder_backward(dout, saved, ew: EncoderW, segs, alpha: float, on_layer_done=None, beta: float = 0.0,
                     ws_tag="bwd_tmp"):
    """dout [M, D] bf16: gradient of the encoder output rows.  Pixels need no gradient.  Parameter gradients are written
    as alpha * grad + beta * old (beta = 1 accumulates micro-batches)."""
    tok, saved_blocks, xl, mean, rstd = saved
    # dx of the final norm's backward is the dY of the last block's fc2: with the transpose-free route its bias gradient (the
    # column sums of dx) comes out of this pass like every other block's (round 4: no stand-alone column sum left in a trunk)
    last_gb = ew.blocks[-1].fc2.gb if _tn_ok(8, 8) else None
    dx = ops.layernorm_bwd(dout, xl, ew.norm.g, mean, rstd, ew.norm.gg, ew.norm.gb, alpha=alpha, accumulate=beta != 0.0,
                           dxsum=last_gb)
    if isinstance(saved_blocks, chain.TrunkCtx):
        side = side_stream(dout.device)
        dx = chain.blocks_backward(dx, saved_blocks, ew, alpha, side.stream.cuda_stream if side.enabled else None,
                                   (lambda li: on_layer_done("enc", li)) if on_layer_done is not None else None,
                                   beta_acc=beta, tag=ws_tag, last_fc2_bias_done=last_gb is not None)
    else:
        nb = len(ew.blocks)
        for li in range(nb - 1, -1, -1):
            dx = block_backward(dx, saved_blocks[li], ew.blocks[li], segs, ew.heads, alpha, beta,
                                fc2_bias_done=li + 1 < nb or last_gb is not None,
                                prev_fc2_gb=ew.blocks[li - 1].fc2.gb if li > 0 else None)
            saved_blocks[li] = None
            if on_layer_done is not None:
                on_layer_done("enc", li)      # bucket launch waits on the side stream's event, not on this stream
    _linear_backward(dx, tok, ew.patch, alpha, need_dx=False, beta=beta)
    side_stream(dout.device).join()
    if on_layer_done is not None:
        on_layer_done("enc", -1)


# =============================================================================================== predictor
def predictor_forward(pw: PredictorW, z, enc_segs: List[Seg], masks_enc, masks_pred, save: bool, ws_tag=None,
                      gemm_flags=0, gates=None):
    """z [sum_i B*Ke_i, D] bf16 (context-encoder output rows, per-mask segments enc_segs).
    Returns (zhat [sum_i B*Kp_i, D] bf16, tgt_segs, saved).  gates: [(0, event)] -- the predictor's weights are one range."""
    Dp = pw.embed.w.shape[0]
    dev = z.device
    _wait_gates(gates, False)
    e = ops.gemm_nt(z, pw.embed.w, bias=pw.embed.b)
    n_tok = len(pw.mask_tokens)
    segs, tsegs, r, rt = [], [], 0, 0
    for sg, mp in zip(enc_segs, masks_pred):
        Kp = mp.shape[1]
        segs.append(Seg(r, sg.B, sg.S + Kp))
        tsegs.append(Seg(rt, sg.B, Kp))
        r += sg.B * (sg.S + Kp)
        rt += sg.B * Kp
    x = torch.empty((r, Dp), dtype=torch.bfloat16, device=dev)
    for i, (sg, psg) in enumerate(zip(enc_segs, segs)):
        # mask_index = i % num_mask_tokens (predictor.py:206; PredictorMultiMaskWrapper passes mask_index=i)
        ops.pred_assemble(_rows(e, sg), pw.mask_tokens[i % n_tok], pw.pos, masks_enc[i], masks_pred[i],
                          out=_rows(x, psg))
    if ws_tag is not None and USE_C_CHAIN:
        x, saved_blocks = chain.blocks_forward(x, pw, segs, save, ws_tag, LN_EPS, gemm_flags=gemm_flags)
    else:
        saved_blocks = []
        for bw in pw.blocks:
            x, sv = block_forward(x, bw, segs, pw.heads, save)
            saved_blocks.append(sv)
    # predictor_norm is row-wise and only target rows are projected (x[:, N_ctxt:], predictor.py:233-237):
    # normalise just those rows.
    t = torch.empty((rt, Dp), dtype=torch.bfloat16, device=dev)
    for sg, psg, tsg in zip(enc_segs, segs, tsegs):
        ops.copy_rows(_rows(x, psg), _rows(t, tsg), psg.B, psg.S, sg.S, tsg.S, 0, tsg.S, Dp)
    tn, mean, rstd = ops.layernorm_fwd(t, pw.norm.g, pw.norm.b, LN_EPS, save_stats=save)
    zhat = ops.gemm_nt(tn, pw.proj.w, bias=pw.proj.b)
    saved = (z, e.shape, segs, tsegs, saved_blocks, t, tn, mean, rstd) if save else None
    return zhat, tsegs, saved


def predictor_backward(dzhat, saved, pw: PredictorW, enc_segs, alpha: float, on_layer_done=None, beta: float = 0.0,
                       ws_tag="bwd_tmp"):
    """dzhat [sum_i B*Kp_i, D] bf16 -> returns dz [sum_i B*Ke_i, D] bf16 (gradient of the encoder output)."""
    z, e_shape, segs, tsegs, saved_blocks, t, tn, mean, rstd = saved
    Dp = pw.embed.w.shape[0]
    n_tok = len(pw.mask_tokens)
    acc = beta != 0.0
    dtn = _linear_backward(dzhat, tn, pw.proj, alpha, beta=beta)
    # the trunk's output gradient is dt on the target rows and zero on the context rows, so the last block's fc2 bias gradient
    # (column sums over ALL rows of that gradient) is the column sum of dt: it comes out of this LayerNorm backward
    last_gb = pw.blocks[-1].fc2.gb if _tn_ok(8, 8) else None
    dt = ops.layernorm_bwd(dtn, t, pw.norm.g, mean, rstd, pw.norm.gg, pw.norm.gb, alpha=alpha, accumulate=acc, dxsum=last_gb)
    total = segs[-1].row0 + segs[-1].rows
    dx = torch.empty((total, Dp), dtype=torch.bfloat16, device=dzhat.device)
    for sg, psg, tsg in zip(enc_segs, segs, tsegs):
        ops.copy_rows(None, _rows(dx, psg), psg.B, 0, 0, psg.S, 0, sg.S, Dp)   # context rows start at zero grad (no ATen fill on the step)
        ops.copy_rows(_rows(dt, tsg), _rows(dx, psg), psg.B, tsg.S, 0, psg.S, sg.S, tsg.S, Dp)
    if on_layer_done is not None:
        on_layer_done("pred", len(pw.blocks))
    if isinstance(saved_blocks, chain.TrunkCtx):
        side = side_stream(dzhat.device)
        dx = chain.blocks_backward(dx, saved_blocks, pw, alpha, side.stream.cuda_stream if side.enabled else None,
                                   (lambda li: on_layer_done("pred", li)) if on_layer_done is not None else None,
                                   beta_acc=beta, tag=ws_tag, last_fc2_bias_done=last_gb is not None)
    else:
        nb = len(pw.blocks)
        for li in range(nb - 1, -1, -1):
            dx = block_backward(dx, saved_blocks[li], pw.blocks[li], segs, pw.heads, alpha, beta,
                                fc2_bias_done=li + 1 < nb or last_gb is not None,
                                prev_fc2_gb=pw.blocks[li - 1].fc2.gb if li > 0 else None)
            saved_blocks[li] = None
            if on_layer_done is not None:
                on_layer_done("pred", li)
    # token assembly backward: mask-token grads = sum of the target rows; context rows flow to predictor_embed
    de = torch.empty(e_shape, dtype=torch.bfloat16, device=dzhat.device)
    used = set()
    for i, (sg, psg) in enumerate(zip(enc_segs, segs)):
        ti = i % n_tok
        ops.colsum(_rows(dx, psg), pw.g_mask_tokens[ti], alpha=alpha, accumulate=(ti in used) or acc, group=psg.S,
                   row_lo=sg.S, row_hi=psg.S)
        used.add(ti)
        ops.copy_rows(_rows(dx, psg), _rows(de, sg), psg.B, psg.S, 0, sg.S, 0, sg.S, Dp)
    for ti in range(n_tok):
        if ti not in used and not acc:
            pw.g_mask_tokens[ti].zero_()
    dz = _linear_backward(de, z, pw.embed, alpha, beta=beta)
    side_stream(dzhat.device).join()
    if on_layer_done is not None:
        on_layer_done("pred", -1)
    return dz
