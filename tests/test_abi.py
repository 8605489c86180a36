"""CPU checks of the drop-in boundary: the C-ABI library builds, loads, and exports exactly what
include/vjepa_hip.h declares (no compute is launched here -- there is no GPU on this host)."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "vjepa_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(vj_[a-z0-9_]+)\s*\(", text)) - {"vj_stream_t"})


def test_library_builds_and_exports_every_declared_symbol():
    from jepa_amd import build
    from jepa_amd.hip import lib as L
    build.build(verbose=False)
    lib = L.load_library()
    names = _declared()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/vjepa_hip.h but not exported"
        assert n in L.SIGNATURES, f"{n} has no ctypes signature"
    for n in L.SIGNATURES:
        assert n in names, f"{n} bound in python but not declared in the header"
    assert lib.vj_abi_version() == 1


def test_no_cpu_fallback_in_ops():
    """The product path must fail loudly on CPU tensors instead of silently computing somewhere else."""
    import pytest
    import torch
    from jepa_amd.hip import ops
    x = torch.zeros(4, 8, dtype=torch.bfloat16)
    with pytest.raises(ValueError):
        ops.layernorm_fwd(x, torch.ones(8), torch.zeros(8), 1e-6)
    with pytest.raises(ValueError):
        ops.gemm_nt(x, x)


def test_chain_workspace_queries_and_host_side_validation():
    """Pure host functions of the launch chains, callable without a GPU: workspace sizes follow the documented layout
    (16*D bf16 per token and block + row statistics + lse; one set + one x buffer without saving), and argument errors are
    rejected before any HIP call (negative return code + message)."""
    import ctypes
    from jepa_amd.hip import lib as L
    lib = L.load_library()
    M, D, H, n = 10560, 1024, 16, 24
    save = lib.vj_blocks_fwd_ws_bytes(M, D, 4 * D, H, n, 1)
    nosave = lib.vj_blocks_fwd_ws_bytes(M, D, 4 * D, H, n, 0)
    per_block = 16 * D * 2 * M + 4 * 4 * M + 4 * H * M + 8 * M   # activations + mean/rstd x2 + lse + the folded LayerNorm's row statistics
    assert save % 256 == 0 and per_block * n <= save <= per_block * n + n * 14 * 256
    assert per_block + 2 * D * M <= nosave <= per_block + 2 * D * M + 14 * 256
    assert lib.vj_blocks_fwd_ws_bytes(2 * M, D, 4 * D, H, n, 1) >= 2 * save - n * 13 * 256
    bwd = lib.vj_blocks_bwd_ws_bytes(M, D, 4 * D, H)
    assert bwd > 2 * (4 * D + D + 3 * D) * 2 * M                 # two parities of du / dx1 / dqkv at least
    assert lib.vj_grad_stats_chunks() == 8 and lib.vj_comm_unique_id_bytes() == 128
    # validation happens on the host, before anything touches a device
    seg = (L.VjSeg * 1)(L.VjSeg(0, 1, 8))
    rc = lib.vj_blocks_fwd(None, 0, None, None, 8, 64, 2, seg, 1, 1e-6, 0, 0, None, 0, None)
    assert rc < 0 and b"no blocks" in lib.vj_last_error()
    blk = (L.VjBlock * 1)()
    blk[0].qkv.n_out, blk[0].qkv.k_in = 3 * 64, 64
    blk[0].proj.n_out, blk[0].proj.k_in = 64, 64
    blk[0].fc1.n_out, blk[0].fc1.k_in = 256, 64
    blk[0].fc2.n_out, blk[0].fc2.k_in = 64, 256
    rc = lib.vj_blocks_fwd(blk, 1, None, None, 8, 64, 2, (L.VjSeg * 1)(L.VjSeg(0, 1, 7)), 1, 1e-6, 0, 0, None, 0, None)
    assert rc < 0 and b"segments" in lib.vj_last_error()
    blk[0].fc2.k_in = 128
    rc = lib.vj_blocks_fwd(blk, 1, None, None, 8, 64, 2, seg, 1, 1e-6, 0, 0, None, 0, None)
    assert rc < 0 and b"inconsistent Linear shapes" in lib.vj_last_error()
    comm = ctypes.c_void_p()
    assert lib.vj_comm_allreduce_bucket(None, None, 4, None) != 0   # null communicator (or no RCCL on this host): refused
