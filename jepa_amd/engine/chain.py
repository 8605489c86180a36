"""Host side of the whole-trunk launch chains (`vj_blocks_fwd` / `vj_blocks_bwd`, include/vjepa_hip.h).

One C call enqueues every kernel of N transformer blocks, so the Python interpreter is out of the launch path of the
~1800 launches of a step.  This module only (i) turns the per-layer weight views of `weights.py` into the C
descriptor array once, (ii) owns the caller-side workspaces (saved activations, backward temporaries) and
(iii) forwards the per-layer "gradients of layer l are enqueued" callback to the gradient reducer.

Reference loops replaced: `for blk in self.blocks: x = blk(x, mask=masks)` (src/models/vision_transformer.py:181-184),
`for blk in self.predictor_blocks: x = blk(x, mask=masks)` (src/models/predictor.py:231-232) and their autograd graph.
"""
import ctypes
import os

import torch

from ..hip.lib import LAYER_CB, VjBlock, VjLinear, VjLnFold, VjNorm, VjSeg, check, get_option, load_library


def _p(t):
    return None if t is None else t.data_ptr()


def _lin(lw):
    w = lw.w
    return VjLinear(_p(w), _p(lw.b), _p(lw.wT), 0 if lw.wT is None else lw.wT.stride(0), _p(lw.gw), _p(lw.gb),
                    w.shape[0], w.shape[1])


def _norm(nw):
    return VjNorm(_p(nw.g), _p(nw.b), _p(nw.gg), _p(nw.gb))


def block_array(blocks):
    """ctypes array of vj_block_t over the (stable) arena views of a list of BlockW; cached on the list object's owner."""
    arr = (VjBlock * len(blocks))()
    for i, b in enumerate(blocks):
        arr[i] = VjBlock(_norm(b.norm1), _lin(b.qkv), _lin(b.proj), _norm(b.norm2), _lin(b.fc1), _lin(b.fc2))
    return arr


def fold_array(folds):
    """ctypes array of vj_lnfold_t over a list of FoldW (stable buffers: refreshed in place)."""
    arr = (VjLnFold * len(folds))()
    for i, f in enumerate(folds):
        arr[i] = VjLnFold(_p(f.w_qkv), _p(f.c_qkv), _p(f.b_qkv), _p(f.w_fc1), _p(f.c_fc1), _p(f.b_fc1))
    return arr


def seg_array(segs):
    arr = (VjSeg * len(segs))()
    for i, s in enumerate(segs):
        arr[i] = VjSeg(s.row0, s.B, s.S)
    return arr


class Workspace:
    """Growable device buffers keyed by tag.  A buffer is only ever replaced after a device-wide synchronise (growth is
    rare: the first steps of a run), so no stream can still be using the old one."""

    _bufs = {}

    @classmethod
    def get(cls, tag, nbytes, device):
        key = (device.index if device.index is not None else torch.cuda.current_device(), tag)
        buf = cls._bufs.get(key)
        if buf is None or buf.numel() < nbytes:
            if buf is not None:
                torch.cuda.synchronize(device)
                cls._bufs[key] = buf = None
            # 30 % headroom: sequence lengths change with every mask draw, and every regrowth costs a device-wide sync
            buf = torch.empty(int(nbytes * 1.3) + 256, dtype=torch.uint8, device=device)
            cls._bufs[key] = buf
        return buf


    @classmethod
    def release(cls, prefix):
        """Drop every buffer whose tag starts with `prefix` (a Trainer's namespace) -- called when the owner dies."""
        for key in [k for k in cls._bufs if str(k[1]).startswith(prefix)]:
            del cls._bufs[key]


class TrunkCtx:
    """What `blocks_backward` needs from `blocks_forward`."""
    __slots__ = ("x_in", "ws", "M", "D", "segs", "seg_arr", "qpre")

    def __init__(self, x_in, ws, M, D, segs, seg_arr, qpre):
        self.x_in, self.ws, self.M, self.D, self.segs, self.seg_arr = x_in, ws, M, D, segs, seg_arr
        self.qpre = qpre   # the forward stored q pre-scaled (option attn_softmax = 2 at forward time): vj_blocks_bwd flags bits 2-3


def _blocks_of(views):
    arr = getattr(views, "_cblocks", None)
    if arr is None:
        arr = block_array(views.blocks)
        views._cblocks = arr
    return arr


def blocks_forward(x, views, segs, save, tag, ln_eps, stream=None, gemm_flags=0, gates=None):
    """x [M, D] bf16 -> (x_out [M, D] bf16, TrunkCtx or None).  `views`: EncoderW / PredictorW (`.blocks`, `.heads`).

    gates: optional [(first_block, torch.cuda.Event)] -- the weights of blocks >= first_block may only be read after the event
    (the fused AdamW / EMA update of the previous step, issued range by range on its own stream: engine/step.py).  The trunk is
    then enqueued as one vj_blocks_fwd call per gated range, each behind a wait on its event; the ranges share the workspace
    exactly as one call lays it out (a range's output IS the next block's saved input slot), so the backward and the results
    are the same as for a single call."""
    lib = load_library()
    M, D = x.shape
    arr = _blocks_of(views)
    n = len(views.blocks)
    Dh = views.blocks[0].fc1.w.shape[0]
    nws = lib.vj_blocks_fwd_ws_bytes(M, D, Dh, views.heads, n, int(save))
    ws = Workspace.get(tag, nws, x.device)
    out = torch.empty_like(x)
    sa = seg_array(segs)
    st = torch.cuda.current_stream().cuda_stream if stream is None else stream
    qpre = get_option("attn_softmax") == 2 and D % 4 == 0   # what vj_blocks_fwd is about to do (chain.hip: (3 * D) % 12 == 0)
    cuts = sorted({b for b, _ in gates if 0 < b < n}) if gates else []
    # folded LayerNorms (vj_blocks_fwd_lnfold): only for a trunk that keeps nothing for a backward
    farr = None
    if getattr(views, "folds", None) is not None and not save:
        farr = getattr(views, "_cfolds", None)
        if farr is None:
            farr = views._cfolds = fold_array(views.folds)

    def call(sub, fsub, nb, xin, xout, flags, ws_ptr, ws_left):
        if fsub is not None:
            check(lib.vj_blocks_fwd_lnfold(sub, fsub, nb, xin, xout, M, D, views.heads, sa, len(segs), ln_eps, flags, ws_ptr, ws_left,
                                           st), "vj_blocks_fwd_lnfold")
        else:
            check(lib.vj_blocks_fwd(sub, nb, xin, xout, M, D, views.heads, sa, len(segs), ln_eps, int(save), flags, ws_ptr, ws_left,
                                    st), "vj_blocks_fwd")
    if not cuts:
        call(arr, farr, n, x.data_ptr(), out.data_ptr(), gemm_flags, ws.data_ptr(), ws.numel())
        return out, (TrunkCtx(x, ws, M, D, segs, sa, qpre) if save else None)
    if stream is not None:
        raise ValueError("blocks_forward: gated ranges are enqueued on the current torch stream")
    ev = {b: e for b, e in gates}
    per_block = lib.vj_blocks_fwd_ws_bytes(M, D, Dh, views.heads, 1, 1)   # one block's saved set; its first member is the block input
    sel_from = (gemm_flags >> 16) & 0xff
    cur = torch.cuda.current_stream()
    cur_in, keep = x.data_ptr(), []
    for b0, b1 in zip([0] + cuts, cuts + [n]):
        if b0 > 0:
            cur.wait_event(ev[b0])
        last = b1 == n
        if save:
            ws_ptr, ws_left = ws.data_ptr() + b0 * per_block, ws.numel() - b0 * per_block
            out_ptr = out.data_ptr() if last else ws.data_ptr() + b1 * per_block
        else:
            ws_ptr, ws_left = ws.data_ptr(), ws.numel()
            if last:
                out_ptr = out.data_ptr()
            else:
                keep.append(torch.empty_like(x))
                out_ptr = keep[-1].data_ptr()
        rel = max(0, sel_from - b0)                       # "first block the kernel selection applies to", relative to this call
        flags = 0 if (gemm_flags >> 16 and rel >= b1 - b0) else (gemm_flags & 0xffff) | (rel << 16)
        sub = ctypes.cast(ctypes.byref(arr, b0 * ctypes.sizeof(VjBlock)), ctypes.POINTER(VjBlock))
        fsub = None if farr is None else ctypes.cast(ctypes.byref(farr, b0 * ctypes.sizeof(VjLnFold)), ctypes.POINTER(VjLnFold))
        call(sub, fsub, b1 - b0, cur_in, out_ptr, flags, ws_ptr, ws_left)
        cur_in = out_ptr
    return out, (TrunkCtx(x, ws, M, D, segs, sa, qpre) if save else None)


def blocks_backward(dout, ctx, views, alpha, side_stream=None, on_layer_done=None, beta_acc=0.0, tag="bwd_tmp",
                    last_fc2_bias_done=False):
    """dout [M, D] bf16 (gradient of the trunk output) -> dx [M, D] bf16; parameter gradients go into the arena views.
    side_stream: raw hipStream_t (int) for the weight gradients or None; on_layer_done(layer) is called per block.
    last_fc2_bias_done: the caller's LayerNorm backward that produced `dout` already wrote the last block's fc2 bias gradient
    (the column sums of dout) -- flags bit 1 of vj_blocks_bwd."""
    lib = load_library()
    M, D = ctx.M, ctx.D
    arr = _blocks_of(views)
    n = len(views.blocks)
    Dh = views.blocks[0].fc1.w.shape[0]
    nws = lib.vj_blocks_bwd_ws_bytes(M, D, Dh, views.heads)
    tmp = Workspace.get(tag, nws, dout.device)
    dx = torch.empty_like(dout)
    errs = []

    def _cb(user, layer):            # runs on this thread, inside vj_blocks_bwd; exceptions must not cross the C frame
        try:
            on_layer_done(layer)
        except BaseException as e:   # noqa: BLE001 -- re-raised right after the call returns
            errs.append(e)

    cb = LAYER_CB(_cb) if on_layer_done is not None else LAYER_CB()
    if side_stream is not None:
        dout.record_stream(torch.cuda.ExternalStream(side_stream, device=dout.device))
    check(lib.vj_blocks_bwd(arr, n, ctx.x_in.data_ptr(), dout.data_ptr(), dx.data_ptr(), M, D, views.heads, ctx.seg_arr,
                            len(ctx.segs), alpha, beta_acc, ctx.ws.data_ptr(), ctx.ws.numel(), tmp.data_ptr(),
                            tmp.numel(), (2 if last_fc2_bias_done else 0) | 8 | (4 if ctx.qpre else 0),
                            torch.cuda.current_stream().cuda_stream, side_stream, cb,
                            None), "vj_blocks_bwd")
    if errs:
        raise errs[0]
    return dx


# ------------------------------------------------------------------------------------------------ profiler front end
def prof_enable(on):
    check(load_library().vj_prof_enable(int(on)), "vj_prof_enable")


def prof_collect(csv_path=None):
    """{family: {launches, ms, flop}} for gemm_nt / attn_fwd / attn_bwd; events are synchronised inside."""
    ms = (ctypes.c_double * 3)()
    fl = (ctypes.c_double * 3)()
    n = (ctypes.c_int64 * 3)()
    check(load_library().vj_prof_collect(ms, fl, n, csv_path.encode() if csv_path else None), "vj_prof_collect")
    return {name: dict(launches=int(n[i]), ms=float(ms[i]), flop=float(fl[i]))
            for i, name in enumerate(("gemm_nt", "attn_fwd", "attn_bwd"))}
