"""Tensor-level wrappers over the C ABI (include/vjepa_hip.h).

Every function takes CUDA(HIP) torch tensors, validates dtype/contiguity, and enqueues the kernel on
`torch.cuda.current_stream()` (or the `stream=` handle given).  PyTorch is only the allocator and stream
provider here; no torch operator computes anything on this path.
"""
import ctypes

import torch

from .lib import check, load_library

BF16 = torch.bfloat16
F32 = torch.float32
I64 = torch.int64

EPI_BF16, EPI_GELU, EPI_DGELU, EPI_F32 = 0, 1, 2, 3
EPI_QKV = 4   # EPI_BF16 whose first N/3 output columns (the q part of a qkv projection) are multiplied by alpha before the rounding


_raw_stream = torch._C._cuda_getCurrentRawStream   # C accessor: ~20x cheaper than torch.cuda.current_stream()
_dev_index = None


def _stream(stream=None):
    global _dev_index
    if stream is not None:
        return ctypes.c_void_p(stream)
    if _dev_index is None:
        _dev_index = torch.cuda.current_device()   # one process per GPU: the device never changes afterwards
    return ctypes.c_void_p(_raw_stream(_dev_index))


def _ptr(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _req(t, dtype, name):
    if t.dtype != dtype:
        raise TypeError(f"{name}: expected {dtype}, got {t.dtype}")
    if not t.is_cuda:
        raise ValueError(f"{name}: must live on the GPU (jepa_amd has no CPU compute path)")
    if not t.is_contiguous():
        raise ValueError(f"{name}: must be contiguous")
    return t


class Scratch:
    """Per-(device, tag, STREAM) scratch workspace for kernels that need partial-sum buffers.  The step runs kernels
    on two streams at once (dgrad chain / weight gradients), so two launches of the same kernel family may be in
    flight together: each stream gets its own buffer.  A buffer is only replaced (regrown) after a device-wide
    synchronise, so no enqueued kernel can still be using the old allocation."""

    _bufs = {}

    @classmethod
    def get(cls, nbytes, device, tag="default", stream=None):
        global _dev_index
        if _dev_index is None:
            _dev_index = torch.cuda.current_device()
        key = (_dev_index, tag, _raw_stream(_dev_index) if stream is None else stream)
        buf = cls._bufs.get(key)
        if buf is None or buf.numel() < nbytes:
            if buf is not None:
                torch.cuda.synchronize()
            buf = torch.empty(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=device)
            cls._bufs[key] = buf
        return buf


# ---------------------------------------------------------------- rows
def gather_rows(src, idx, out=None, stream=None):
    """out[b,k,:] = src[b, idx[b,k], :]; src [B,N,D] or a broadcast table [1,N,D] / [N,D].  Bit-exact."""
    lib = load_library()
    idx = _req(idx, I64, "idx")
    B, K = idx.shape
    if not src.is_contiguous():
        raise ValueError("src must be contiguous")
    D = src.shape[-1]
    if src.dim() == 2 or (src.shape[0] == 1 and B > 1):
        bstride = 0  # broadcast table (pos-embed)
    elif src.dim() == 3 and src.shape[0] == B:
        bstride = src.shape[1]
    else:
        raise ValueError(f"gather_rows: batch mismatch src {tuple(src.shape)} idx {tuple(idx.shape)}")
    if out is None:
        out = torch.empty((B, K, D), dtype=src.dtype, device=src.device)
    check(lib.vj_gather_rows(_ptr(src), _ptr(out), _ptr(idx), B, K, D * src.element_size(), bstride, _stream(stream)),
          "vj_gather_rows")
    return out


def scatter_rows(src, idx, N, stream=None):
    """zeros[B,N,D] with out[b, idx[b,k], :] = src[b,k,:] (backward of gather_rows)."""
    lib = load_library()
    idx = _req(idx, I64, "idx")
    B, K = idx.shape
    D = src.shape[-1]
    out = torch.empty((B, N, D), dtype=src.dtype, device=src.device)
    check(lib.vj_scatter_rows(_ptr(src.contiguous()), _ptr(out), _ptr(idx), B, N, K, D * src.element_size(),
                              _stream(stream)), "vj_scatter_rows")
    return out


def copy_rows(src, dst, B, src_rows, src_off, dst_rows, dst_off, n, D, stream=None):
    lib = load_library()
    check(lib.vj_copy_rows(_ptr(src), _ptr(dst), B, src_rows, src_off, dst_rows, dst_off, n, D, _stream(stream)),
          "vj_copy_rows")
    return dst


def tubelet_pack(clips, tubelet, patch, idx=None, out=None, stream=None):
    lib = load_library()
    clips = _req(clips, F32, "clips")
    B, C, T, H, W = clips.shape
    N = (T // tubelet) * (H // patch) * (W // patch)
    K = N if idx is None else idx.shape[1]
    kdim = C * tubelet * patch * patch
    if out is None:
        out = torch.empty((B * K, kdim), dtype=BF16, device=clips.device)
    check(lib.vj_tubelet_pack(_ptr(clips), _ptr(out), _ptr(idx), B, C, T, H, W, tubelet, patch, K, _stream(stream)),
          "vj_tubelet_pack")
    return out


def add_pos(x, pos, B, K, idx=None, stream=None):
    """x [B*K, D] bf16 += pos[(idx or arange)[b,k]] (fp32 table [N,D])."""
    lib = load_library()
    _req(x, BF16, "x")
    _req(pos, F32, "pos")
    check(lib.vj_add_pos(_ptr(x), _ptr(pos), _ptr(idx), B, K, x.shape[-1], _stream(stream)), "vj_add_pos")
    return x


# ---------------------------------------------------------------- layernorm
def layernorm_fwd(x, gamma, beta, eps, save_stats=True, out=None, stream=None):
    lib = load_library()
    _req(x, BF16, "x")
    rows, D = x.numel() // x.shape[-1], x.shape[-1]
    y = torch.empty_like(x) if out is None else out
    mean = rstd = None
    if save_stats:
        mean = torch.empty(rows, dtype=F32, device=x.device)
        rstd = torch.empty(rows, dtype=F32, device=x.device)
    check(lib.vj_layernorm_fwd(_ptr(x), _ptr(gamma), _ptr(beta), _ptr(y), _ptr(mean), _ptr(rstd), rows, D, eps,
                               _stream(stream)), "vj_layernorm_fwd")
    return y, mean, rstd


def layernorm_bwd(dy, x, gamma, mean, rstd, dgamma, dbeta, dres=None, alpha=1.0, accumulate=False, dxsum=None,
                  stream=None):
    """dxsum (optional fp32 [D]): also alpha * column sums of the returned dx (+ old when accumulating) -- the bias
    gradient of the Linear whose dY this dx is (include/vjepa_hip.h: vj_layernorm_bwd_colsum)."""
    lib = load_library()
    _req(dy, BF16, "dy")
    rows, D = x.numel() // x.shape[-1], x.shape[-1]
    dx = torch.empty_like(x)
    nws = lib.vj_layernorm_bwd_ws_bytes(D)
    ws = Scratch.get(nws, x.device, "ln", stream=stream)
    check(lib.vj_layernorm_bwd_colsum(_ptr(dy), _ptr(x), _ptr(gamma), _ptr(mean), _ptr(rstd), _ptr(dres), _ptr(dx),
                                      _ptr(dgamma), _ptr(dbeta), _ptr(dxsum), alpha, 1.0 if accumulate else 0.0, rows, D,
                                      _ptr(ws), nws, _stream(stream)), "vj_layernorm_bwd")
    return dx


# ---------------------------------------------------------------- gemm
GEMM_FLAGS = 0  # default kernel-selection flags of vj_gemm_bf16_nt (0 = automatic; gemm.hip dispatch_gemm)
KERNEL_TIMERS = None  # bench.py sets this to a dict: name -> list of (start_event, end_event, work) on the launch stream


def _timed(name, work):
    """Context helper: HIP events around one launch on the stream the kernel is enqueued on."""
    if KERNEL_TIMERS is None:
        return None
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    KERNEL_TIMERS.setdefault(name, []).append((s, e, work))
    s.record()
    return e


def gemm_nt(A, B, out=None, bias=None, residual=None, aux_in=None, aux_out=None, epilogue=EPI_BF16, alpha=1.0,
            beta=0.0, M=None, K=None, flags=None, stream=None):
    """C[M,N] = A[M,K] @ B[N,K]^T with a fused epilogue.  A,B bf16 2-D (row stride may exceed K)."""
    lib = load_library()
    for t, nm in ((A, "A"), (B, "B")):
        if t.dtype != BF16 or not t.is_cuda or t.dim() != 2 or t.stride(1) != 1:
            raise ValueError(f"gemm_nt: {nm} must be a 2-D bf16 GPU tensor with unit inner stride")
    M = A.shape[0] if M is None else M
    K = A.shape[1] if K is None else K
    N = B.shape[0]
    if out is None:
        out = torch.empty((M, N), dtype=F32 if epilogue == EPI_F32 else BF16, device=A.device)
    aux = aux_in if aux_in is not None else aux_out
    _ev = _timed("gemm_nt", 2.0 * M * N * K)
    check(lib.vj_gemm_bf16_nt(_ptr(A), A.stride(0), _ptr(B), B.stride(0), _ptr(out), out.stride(0), M, N, K,
                              _ptr(bias), _ptr(residual), 0 if residual is None else residual.stride(0),
                              _ptr(aux_in), _ptr(aux_out), 0 if aux is None else aux.stride(0), epilogue, alpha, beta,
                              GEMM_FLAGS if flags is None else flags, _stream(stream)), "vj_gemm_bf16_nt")
    if _ev is not None:
        _ev.record()
    return out


def ln_rowstats(x, eps, out=None, stream=None):
    """rowstats[m] = {rstd_m, -mean_m * rstd_m} (fp32 [M, 2]) of the bf16 rows x [M, D]: the statistics pass of a LayerNorm that is
    folded into the GEMM consuming it (vj_ln_rowstats)."""
    lib = load_library()
    _req(x, BF16, "x")
    M, D = x.shape
    rs = torch.empty((M, 2), dtype=F32, device=x.device) if out is None else out
    check(lib.vj_ln_rowstats(_ptr(x), _ptr(rs), M, D, eps, _stream(stream)), "vj_ln_rowstats")
    return rs


def ln_fold_weights(W, b, gamma, beta, Wf=None, cvec=None, bf=None, stream=None):
    """Wf = bf16(W * gamma) [N, K], cvec[n] = sum_k Wf[n, k], bf = b + W beta (vj_ln_fold_weights); W fp32 [N, K] contiguous."""
    lib = load_library()
    for t, nm in ((W, "W"), (gamma, "gamma"), (beta, "beta")):
        if t.dtype != F32 or not t.is_cuda or not t.is_contiguous():
            raise ValueError(f"ln_fold_weights: {nm} must be a contiguous fp32 GPU tensor")
    N, K = W.shape
    Wf = torch.empty((N, K), dtype=BF16, device=W.device) if Wf is None else Wf
    cvec = torch.empty((N,), dtype=F32, device=W.device) if cvec is None else cvec
    bf = torch.empty((N,), dtype=F32, device=W.device) if bf is None else bf
    check(lib.vj_ln_fold_weights(_ptr(W), _ptr(b), _ptr(gamma), _ptr(beta), _ptr(Wf), _ptr(cvec), _ptr(bf), N, K, _stream(stream)),
          "vj_ln_fold_weights")
    return Wf, cvec, bf


def gemm_nt_lnfold(X, Wf, bf, rowstats, cvec, out=None, epilogue=EPI_BF16, alpha=1.0, flags=None, stream=None):
    """out = LayerNorm(X) W^T + b from the RAW bf16 rows X [M, K] (vj_gemm_bf16_nt_lnfold): Wf / cvec / bf from ln_fold_weights,
    rowstats from ln_rowstats; epilogue EPI_BF16, EPI_GELU or EPI_QKV (first N/3 columns times alpha)."""
    lib = load_library()
    for t, nm in ((X, "X"), (Wf, "Wf")):
        if t.dtype != BF16 or not t.is_cuda or t.dim() != 2 or t.stride(1) != 1:
            raise ValueError(f"gemm_nt_lnfold: {nm} must be a 2-D bf16 GPU tensor with unit inner stride")
    M, K = X.shape
    N = Wf.shape[0]
    if out is None:
        out = torch.empty((M, N), dtype=BF16, device=X.device)
    check(lib.vj_gemm_bf16_nt_lnfold(_ptr(X), X.stride(0), _ptr(Wf), Wf.stride(0), _ptr(out), out.stride(0), M, N, K, _ptr(bf),
                                     _ptr(rowstats), _ptr(cvec), epilogue, alpha, GEMM_FLAGS if flags is None else flags,
                                     _stream(stream)), "vj_gemm_bf16_nt_lnfold")
    return out


WGRAD_WS_BYTES = 96 << 20
GROUP_WS_BYTES = 192 << 20   # the grouped launch's workspace, same size as the C chain's (same split factor -> same bits)


def gemm_wgrad(dyT, xT, out, alpha=1.0, beta=0.0, flags=None, stream=None):
    """out[N_out, K_in] (fp32) = alpha * dyT[N_out, Tp] @ xT[K_in, Tp]^T + beta*out, split-K over the tokens."""
    lib = load_library()
    M, K = dyT.shape
    N = xT.shape[0]
    ws = Scratch.get(WGRAD_WS_BYTES, dyT.device, "wgrad", stream=stream)
    _ev = _timed("gemm_nt", 2.0 * M * N * K)
    check(lib.vj_gemm_bf16_nt_splitk(_ptr(dyT), dyT.stride(0), _ptr(xT), xT.stride(0), _ptr(out), out.stride(0), M, N,
                                     K, alpha, beta, GEMM_FLAGS if flags is None else flags, _ptr(ws), WGRAD_WS_BYTES,
                                     _stream(stream)), "vj_gemm_bf16_nt_splitk")
    if _ev is not None:
        _ev.record()
    return out


def gemm_wgrad_tn_grouped(problems, alpha=1.0, beta=0.0, stream=None):
    """problems: [(dy [T, N1] bf16, x [T, N2] bf16, out [N1, N2] fp32), ...] (at most 4, same T): every out = alpha *
    dy^T @ x + beta * out in ONE launch (vj_gemm_bf16_tn_grouped)."""
    lib = load_library()
    T = problems[0][0].shape[0]
    flat = []
    for dy, x, out in problems:
        _req(dy, BF16, "dy")
        _req(x, BF16, "x")
        assert dy.shape[0] == T and x.shape[0] == T and out.shape == (dy.shape[1], x.shape[1]) and out.dtype == F32
        flat += [dy.data_ptr(), dy.stride(0), x.data_ptr(), x.stride(0), out.data_ptr(), out.stride(0), dy.shape[1], x.shape[1]]
    arr = (ctypes.c_int64 * len(flat))(*[int(v) for v in flat])
    ws = Scratch.get(GROUP_WS_BYTES, problems[0][0].device, "wgrad_group", stream=stream)
    _ev = _timed("gemm_nt", sum(2.0 * T * dy.shape[1] * x.shape[1] for dy, x, _ in problems))
    check(lib.vj_gemm_bf16_tn_grouped(ctypes.addressof(arr), len(problems), T, alpha, beta, _ptr(ws), GROUP_WS_BYTES,
                                      _stream(stream)), "vj_gemm_bf16_tn_grouped")
    if _ev is not None:
        _ev.record()


def gemm_wgrad_tn(dy, x, out, alpha=1.0, beta=0.0, stream=None):
    """out[N_out, K_in] (fp32) = alpha * dy[T, N_out]^T @ x[T, K_in] + beta*out: no operand transposes, split-K over T."""
    lib = load_library()
    _req(dy, BF16, "dy")
    _req(x, BF16, "x")
    T, N1 = dy.shape
    N2 = x.shape[1]
    ws = Scratch.get(WGRAD_WS_BYTES, dy.device, "wgrad", stream=stream)
    _ev = _timed("gemm_nt", 2.0 * T * N1 * N2)
    check(lib.vj_gemm_bf16_tn_splitk(_ptr(dy), dy.stride(0), _ptr(x), x.stride(0), _ptr(out), out.stride(0), T, N1, N2,
                                     alpha, beta, _ptr(ws), WGRAD_WS_BYTES, _stream(stream)), "vj_gemm_bf16_tn_splitk")
    if _ev is not None:
        _ev.record()
    return out


def transpose_colsum(x, colsum_out, alpha=1.0, accumulate=False, stream=None):
    """x [M,N] bf16 -> x^T [N, pad64(M)] and colsum_out[n] = alpha*sum_m x[m,n] (+ old) in one pass."""
    lib = load_library()
    _req(x, BF16, "x")
    M, N = x.shape
    Mp = pad64(M)
    out = torch.empty((N, Mp), dtype=BF16, device=x.device)
    nws = lib.vj_transpose_colsum_ws_bytes(M, N)
    ws = Scratch.get(nws, x.device, "tcolsum", stream=stream)
    check(lib.vj_transpose_colsum_bf16(_ptr(x), _ptr(out), M, N, x.stride(0), Mp, _ptr(colsum_out), alpha,
                                       1.0 if accumulate else 0.0, _ptr(ws), nws, _stream(stream)),
          "vj_transpose_colsum_bf16")
    return out


def pad64(m):
    return (m + 63) // 64 * 64


def transpose(x, M=None, out=None, stream=None):
    """x [M,N] bf16 -> [N, pad64(M)] (zero padded): K-contiguous wgrad operand."""
    lib = load_library()
    _req(x, BF16, "x")
    M = x.shape[0] if M is None else M
    N = x.shape[1]
    Mp = pad64(M)
    if out is None:
        out = torch.empty((N, Mp), dtype=BF16, device=x.device)
    check(lib.vj_transpose_bf16(_ptr(x), _ptr(out), M, N, x.stride(0), Mp, _stream(stream)), "vj_transpose_bf16")
    return out


def transpose_multi(desc, blocks, n_blocks, stream=None):
    lib = load_library()
    check(lib.vj_transpose_multi(_ptr(desc), _ptr(blocks), n_blocks, _stream(stream)), "vj_transpose_multi")


def colsum(x, out, M=None, alpha=1.0, accumulate=False, group=0, row_lo=0, row_hi=None, stream=None):
    """out[n] (fp32) = alpha * sum_m x[m,n] (+ out); optional per-sample row window for the mask-token grad."""
    lib = load_library()
    _req(x, BF16, "x")
    M = x.shape[0] if M is None else M
    N = x.shape[1]
    nws = lib.vj_colsum_ws_bytes(N)
    ws = Scratch.get(nws, x.device, "colsum", stream=stream)
    if group <= 0:
        group, row_lo, row_hi = max(M, 1), 0, max(M, 1)
    check(lib.vj_colsum_bf16(_ptr(x), M, N, x.stride(0), group, row_lo, row_hi, _ptr(out), alpha,
                             1.0 if accumulate else 0.0, _ptr(ws), nws, _stream(stream)), "vj_colsum_bf16")
    return out


# ---------------------------------------------------------------- attention
def attn_fwd(qkv, B, S, H, hd, scale, save_lse=True, out=None, stream=None):
    """qkv [B*S, 3*H*hd] bf16 (packed [B,S,3,H,hd]) -> o [B*S, H*hd], lse2 [B,H,S]."""
    lib = load_library()
    _req(qkv, BF16, "qkv")
    o = torch.empty((B * S, H * hd), dtype=BF16, device=qkv.device) if out is None else out
    lse = torch.empty((B, H, S), dtype=F32, device=qkv.device) if save_lse else None
    _ev = _timed("attn_fwd", 4.0 * B * H * S * S * hd)
    check(lib.vj_attn_fwd(_ptr(qkv), _ptr(o), _ptr(lse), B, S, H, hd, scale, _stream(stream)), "vj_attn_fwd")
    if _ev is not None:
        _ev.record()
    return o, lse


def attn_bwd(qkv, o, dout, lse, B, S, H, hd, scale, out=None, stream=None):
    lib = load_library()
    _req(dout, BF16, "dout")
    dqkv = torch.empty_like(qkv) if out is None else out
    nws = lib.vj_attn_bwd_segs_ws_bytes(_seg_array([(0, B, S)]), 1, H, hd)
    ws = Scratch.get(nws, qkv.device, "attn", stream=stream)
    _ev = _timed("attn_bwd", 8.0 * B * H * S * S * hd)
    check(lib.vj_attn_bwd(_ptr(qkv), _ptr(o), _ptr(dout), _ptr(lse), _ptr(dqkv), B, S, H, hd, scale, _ptr(ws), nws,
                          _stream(stream)), "vj_attn_bwd")
    if _ev is not None:
        _ev.record()
    return dqkv


def _seg_array(segs):
    from .lib import VjSeg
    arr = (VjSeg * len(segs))()
    for i, (row0, B, S) in enumerate(segs):
        arr[i] = VjSeg(row0, B, S)
    return arr


def attn_fwd_segs(qkv, segs, H, hd, scale, save_lse=True, stream=None):
    """Attention over several [B_i, S_i] segments of one activation in ONE launch (vj_attn_fwd_segs).  qkv [M, 3*H*hd] bf16;
    segs: list of (row0, B, S) -> o [M, H*hd], lse2 flat [H*M] (segment i: [B_i, H, S_i] at H*row0_i)."""
    lib = load_library()
    _req(qkv, BF16, "qkv")
    M = qkv.shape[0]
    o = torch.empty((M, H * hd), dtype=BF16, device=qkv.device)
    lse = torch.empty((H * M,), dtype=F32, device=qkv.device) if save_lse else None
    check(lib.vj_attn_fwd_segs(_ptr(qkv), _ptr(o), _ptr(lse), _seg_array(segs), len(segs), H, hd, scale, _stream(stream)),
          "vj_attn_fwd_segs")
    return o, lse


def attn_bwd_segs(qkv, o, dout, lse, segs, H, hd, scale, colsum=False, stream=None):
    """Backward of attn_fwd_segs in one launch pair (vj_attn_bwd_segs) -> dqkv [, colq, colkv: the segments' column partials one
    after the other]."""
    import ctypes
    lib = load_library()
    _req(dout, BF16, "dout")
    M = qkv.shape[0]
    dqkv = torch.empty_like(qkv)
    seg_arr = _seg_array(segs)
    nws = lib.vj_attn_bwd_segs_ws_bytes(seg_arr, len(segs), H, hd)
    ws = Scratch.get(nws, qkv.device, "attn", stream=stream)
    colq = colkv = None
    if colsum:
        rq_t = rkv_t = 0
        for _, B, S in segs:
            rq, rkv = ctypes.c_int64(0), ctypes.c_int64(0)
            check(lib.vj_attn_bwd_colsum_rows(B, S, hd, ctypes.byref(rq), ctypes.byref(rkv)), "vj_attn_bwd_colsum_rows")
            rq_t, rkv_t = rq_t + (rq.value if B * S else 0), rkv_t + (rkv.value if B * S else 0)
        colq = torch.empty((rq_t, H * hd), dtype=F32, device=qkv.device)
        colkv = torch.empty((rkv_t, 2 * H * hd), dtype=F32, device=qkv.device)
    check(lib.vj_attn_bwd_segs(_ptr(qkv), _ptr(o), _ptr(dout), _ptr(lse), _ptr(dqkv), seg_arr, len(segs), H, hd, scale,
                               _ptr(ws), nws, _ptr(colq), _ptr(colkv), _stream(stream)), "vj_attn_bwd_segs")
    return (dqkv, colq, colkv) if colsum else dqkv


def attn_bwd_colsum(qkv, o, dout, lse, B, S, H, hd, scale, out=None, stream=None):
    """attn_bwd + fp32 column partials of dqkv over this segment (vj_attn_bwd_colsum): returns (dqkv, colq [rows_q, H*hd],
    colkv [rows_kv, 2*H*hd]); the qkv bias gradient is colq.sum(0) | colkv.sum(0) (reduce_segments on the product path)."""
    import ctypes
    lib = load_library()
    _req(dout, BF16, "dout")
    dqkv = torch.empty_like(qkv) if out is None else out
    rq, rkv = ctypes.c_int64(0), ctypes.c_int64(0)
    check(lib.vj_attn_bwd_colsum_rows(B, S, hd, ctypes.byref(rq), ctypes.byref(rkv)), "vj_attn_bwd_colsum_rows")
    colq = torch.empty((rq.value, H * hd), dtype=F32, device=qkv.device)
    colkv = torch.empty((rkv.value, 2 * H * hd), dtype=F32, device=qkv.device)
    nws = lib.vj_attn_bwd_segs_ws_bytes(_seg_array([(0, B, S)]), 1, H, hd)
    ws = Scratch.get(nws, qkv.device, "attn", stream=stream)
    check(lib.vj_attn_bwd_colsum(_ptr(qkv), _ptr(o), _ptr(dout), _ptr(lse), _ptr(dqkv), B, S, H, hd, scale, _ptr(ws), nws,
                                 _ptr(colq), _ptr(colkv), _stream(stream)), "vj_attn_bwd_colsum")
    return dqkv, colq, colkv


def gemm_dgelu_colsum(dy, wT, aux_in, stream=None):
    """fc2 dgrad with fc1's bias-gradient partials (vj_gemm_bf16_nt_dgelu_colsum): returns (du bf16 [M, N], colpart fp32
    [rows, N] or None when the fused kernel did not take the problem -- du is the plain GEMM's either way)."""
    import ctypes
    lib = load_library()
    _req(dy, BF16, "dy")
    M, K = dy.shape
    N = wT.shape[0]
    du = torch.empty((M, N), dtype=BF16, device=dy.device)
    rows = lib.vj_gemm_colsum_rows(M)
    colpart = torch.empty((rows, N), dtype=F32, device=dy.device)
    fused = ctypes.c_int(0)
    check(lib.vj_gemm_bf16_nt_dgelu_colsum(_ptr(dy), dy.stride(0), _ptr(wT), wT.stride(0), _ptr(du), N, M, N, K, _ptr(aux_in),
                                           aux_in.stride(0), _ptr(colpart), rows, 0, ctypes.byref(fused), _stream(stream)),
          "vj_gemm_bf16_nt_dgelu_colsum")
    return du, (colpart if fused.value else None)


def reduce_segments(segs, alpha=1.0, accumulate=False, stream=None):
    """segs: list of (part fp32 [P, stride] (a 2-D tensor or a column slice of one), out fp32 [N]) -> out = alpha * column sums
    (+ old when accumulating), all in ONE launch (vj_reduce_segments)."""
    from .lib import VjReduceSeg
    lib = load_library()
    arr = (VjReduceSeg * len(segs))()
    for i, (part, out) in enumerate(segs):
        if part.dtype != F32 or out.dtype != F32 or not part.is_cuda or part.dim() != 2 or part.stride(1) != 1:
            raise ValueError("reduce_segments: partials must be 2-D fp32 GPU tensors with unit column stride")
        arr[i] = VjReduceSeg(part.data_ptr(), out.data_ptr(), part.shape[0], out.numel(), part.stride(0))
    check(lib.vj_reduce_segments(arr, len(segs), alpha, 1.0 if accumulate else 0.0, _stream(stream)), "vj_reduce_segments")


# ---------------------------------------------------------------- predictor / loss
def xattn_fwd(q, kv, B, NQ, N, H, hd, scale, resid=None, shared_q=True, save_lse=True, stream=None):
    """Few-query cross-attention (vj_xattn_fwd): q [NQ, D] (shared_q) or [B, NQ, D], kv [B*N, 2*D] packed -> out [B*NQ, D] bf16,
    lse2 [B, H, NQ] fp32 (None when not saved)."""
    D = H * hd
    _req(q, torch.bfloat16, "q"); _req(kv, torch.bfloat16, "kv")
    out = torch.empty((B * NQ, D), dtype=torch.bfloat16, device=kv.device)
    lse = torch.empty((B, H, NQ), dtype=torch.float32, device=kv.device) if save_lse else None
    check(load_library().vj_xattn_fwd(_ptr(q), 0 if shared_q else NQ * D, _ptr(kv), _ptr(resid), _ptr(out), _ptr(lse), B, NQ, N, H,
                                      hd, float(scale), _stream(stream)), "vj_xattn_fwd")
    return out, lse


def xattn_bwd(q, kv, dy, lse, B, N, H, hd, scale, shared_q=True, stream=None):
    """Backward of xattn_fwd for one query per sample: returns dq [B, D] bf16 (per sample) and dkv [B*N, 2*D] bf16."""
    D = H * hd
    _req(q, torch.bfloat16, "q"); _req(kv, torch.bfloat16, "kv"); _req(dy, torch.bfloat16, "dy")
    dq = torch.empty((B, D), dtype=torch.bfloat16, device=kv.device)
    dkv = torch.empty_like(kv)
    check(load_library().vj_xattn_bwd(_ptr(q), 0 if shared_q else D, _ptr(kv), _ptr(dy), _ptr(lse), _ptr(dq), _ptr(dkv), B, 1, N, H,
                                      hd, float(scale), _stream(stream)), "vj_xattn_bwd")
    return dq, dkv


def pred_assemble(e, mask_token, pos, idx_e, idx_p, out=None, stream=None):
    lib = load_library()
    _req(e, BF16, "e")
    B, Ke = idx_e.shape
    Kp = idx_p.shape[1]
    D = e.shape[-1]
    if out is None:
        out = torch.empty((B * (Ke + Kp), D), dtype=BF16, device=e.device)
    check(lib.vj_pred_assemble_fwd(_ptr(e), _ptr(mask_token), _ptr(pos), _ptr(idx_e), _ptr(idx_p), _ptr(out), B, Ke,
                                   Kp, D, _stream(stream)), "vj_pred_assemble_fwd")
    return out


def target_rows(x, gamma, beta, idx, B, N, eps_norm, eps_ln=1e-5, out=None, stream=None):
    lib = load_library()
    _req(x, BF16, "x")
    idx = _req(idx, I64, "idx")
    K = idx.shape[1]
    D = x.shape[-1]
    h = torch.empty((B, K, D), dtype=F32, device=x.device) if out is None else out
    check(lib.vj_target_rows(_ptr(x), _ptr(gamma), _ptr(beta), _ptr(idx), _ptr(h), B, N, K, D, eps_norm, eps_ln,
                             _stream(stream)), "vj_target_rows")
    return h


def latent_loss(z, h, loss_out, p=1.0, out_scale=1.0, accumulate=False, dz=None, gscale=0.0, stream=None):
    lib = load_library()
    _req(z, BF16, "z")
    _req(h, F32, "h")
    nws = lib.vj_latent_loss_ws_bytes()
    ws = Scratch.get(nws, z.device, "loss", stream=stream)
    check(lib.vj_latent_loss(_ptr(z), _ptr(h), _ptr(dz), z.numel(), p, gscale, out_scale, int(accumulate),
                             _ptr(loss_out), _ptr(ws), nws, _stream(stream)), "vj_latent_loss")
    return loss_out


def token_pstd(z, pstd, B, K, D, accumulate, stats=None, stream=None):
    lib = load_library()
    check(lib.vj_token_pstd(_ptr(z), _ptr(pstd), _ptr(stats), B, K, D, int(accumulate), _stream(stream)),
          "vj_token_pstd")


def reg_grad(z, pstd_sum, stats, dz, B, K, D, n_masks, coef, stream=None):
    lib = load_library()
    check(lib.vj_reg_grad(_ptr(z), _ptr(pstd_sum), _ptr(stats), _ptr(dz), B, K, D, n_masks, coef, _stream(stream)),
          "vj_reg_grad")


def reg_finish(pstd, n_masks, out, stream=None):
    lib = load_library()
    check(lib.vj_reg_finish(_ptr(pstd), pstd.numel(), n_masks, _ptr(out), _stream(stream)), "vj_reg_finish")


# ---------------------------------------------------------------- optimizer
def adamw_ema(p, g, m, v, p_bf16, tgt, tgt_bf16, lr, wd, beta1, beta2, eps, step, gscale, ema, stream=None):
    lib = load_library()
    check(lib.vj_adamw_ema(_ptr(p), _ptr(g), _ptr(m), _ptr(v), _ptr(p_bf16), _ptr(tgt), _ptr(tgt_bf16), p.numel(),
                           lr, wd, beta1, beta2, eps, step, gscale, ema, _stream(stream)), "vj_adamw_ema")


def step_advance(gstat, step_dev, stream=None):
    check(load_library().vj_step_advance(_ptr(gstat), _ptr(step_dev), _stream(stream)), "vj_step_advance")


def adamw_ema_guarded(p, g, m, v, p_bf16, tgt, tgt_bf16, lr, wd, beta1, beta2, eps, gscale, ema, gstat, sel, clip,
                      norm_scale, step_dev, stream=None):
    """AdamW + EMA + bf16 re-casts with the skip-on-non-finite flag, the clip coefficient and the step count read on
    the device (include/vjepa_hip.h: vj_adamw_ema_guarded)."""
    lib = load_library()
    check(lib.vj_adamw_ema_guarded(_ptr(p), _ptr(g), _ptr(m), _ptr(v), _ptr(p_bf16), _ptr(tgt), _ptr(tgt_bf16), p.numel(),
                                   lr, wd, beta1, beta2, eps, gscale, ema, _ptr(gstat), sel, clip, norm_scale,
                                   _ptr(step_dev), _stream(stream)), "vj_adamw_ema_guarded")


def grad_stats_multi(G, M1, M2, desc, n_tensors, stream=None):
    """One launch over the arenas -> fp32 [n_tensors, 3] = per-tensor {sum g^2, sum |exp_avg|, sum |exp_avg_sq|}
    (device tensor; reading it is the only host sync)."""
    lib = load_library()
    nch = lib.vj_grad_stats_chunks()
    out = torch.empty((n_tensors, nch, 3), dtype=F32, device=G.device)
    check(lib.vj_grad_stats_multi(_ptr(G), _ptr(M1), _ptr(M2), _ptr(desc), n_tensors, _ptr(out), _stream(stream)),
          "vj_grad_stats_multi")
    return out.sum(dim=1)


def ema_update(tgt, src, tgt_bf16, m, stream=None):
    lib = load_library()
    check(lib.vj_ema_update(_ptr(tgt), _ptr(src), _ptr(tgt_bf16), tgt.numel(), m, _stream(stream)), "vj_ema_update")


def cast_bf16(src, dst, stream=None):
    lib = load_library()
    check(lib.vj_cast_f32_to_bf16(_ptr(src), _ptr(dst), src.numel(), _stream(stream)), "vj_cast_f32_to_bf16")
    return dst


def sqnorm(g, out2, accumulate=False, stream=None):
    lib = load_library()
    nws = lib.vj_sqnorm_ws_bytes()
    ws = Scratch.get(nws, g.device, "sqnorm", stream=stream)
    check(lib.vj_sqnorm_f32(_ptr(g), g.numel(), _ptr(out2), int(accumulate), _ptr(ws), nws, _stream(stream)),
          "vj_sqnorm_f32")
    return out2
