#!/usr/bin/env python
"""What would ONE grouped weight-gradient launch per transformer block buy?  Upper-bound probe with the existing
kernel: the four weight gradients of a block (qkv, proj, fc1, fc2) as four launches (each with its own split-K and
slice reduction) against a single TN GEMM with the same number of 256x256 output tiles, the same token count and
(ws_cap) a workspace cap that forces split-K off."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from jepa_amd.hip import ops  # noqa: E402
from jepa_amd.hip.lib import load_library  # noqa: E402
from jepa_amd.hip.ops import _ptr, _stream, check  # noqa: E402


def timeit(fn, reps=10):
    for _ in range(2):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3


def tn(dy, x, out, ws, ws_bytes):
    lib = load_library()
    T, N1 = dy.shape
    N2 = x.shape[1]
    check(lib.vj_gemm_bf16_tn_splitk(_ptr(dy), dy.stride(0), _ptr(x), x.stride(0), _ptr(out), out.stride(0), T, N1, N2,
                                     1.0, 0.0, _ptr(ws), ws_bytes, _stream(None)), "tn")


def main():
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(0)
    ws = torch.empty(96 << 20, dtype=torch.uint8, device=dev)
    for tag, T, D, Dh in (("ctx block", 10560, 1024, 4096), ("prd block", 58560, 384, 1536)):
        shapes = [(3 * D, D), (D, D), (Dh, D), (D, Dh)]
        ops_ = []
        for n1, n2 in shapes:
            dy = torch.randn(T, n1, device=dev, generator=g).to(torch.bfloat16)
            x = torch.randn(T, n2, device=dev, generator=g).to(torch.bfloat16)
            out = torch.empty(n1, n2, device=dev, dtype=torch.float32)
            ops_.append((dy, x, out))
        fl = sum(2.0 * T * a * b for a, b in shapes)
        us4 = timeit(lambda: [tn(dy, x, out, ws, 96 << 20) for dy, x, out in ops_])
        each = [timeit(lambda o=o: tn(o[0], o[1], o[2], ws, 96 << 20)) for o in ops_]
        # equivalent single problem: the same output area as one [sum n1*n2 / D, D] matrix
        n1e = int(sum(a * b for a, b in shapes) // D)
        dy = torch.randn(T, n1e, device=dev, generator=g).to(torch.bfloat16)
        x = torch.randn(T, D, device=dev, generator=g).to(torch.bfloat16)
        out = torch.empty(n1e, D, device=dev, dtype=torch.float32)
        us1_free = timeit(lambda: tn(dy, x, out, ws, 96 << 20))
        us1_nosplit = timeit(lambda: tn(dy, x, out, ws, n1e * D * 4))
        print(f"{tag} T={T}: four launches {us4:7.1f} us ({fl / us4 / 1e6:6.0f} TF/s; each " + " ".join(f"{u:.0f}" for u in each) +
              f") | one [{n1e}x{D}] problem, split-K free {us1_free:7.1f} us ({fl / us1_free / 1e6:6.0f} TF/s), "
              f"no split-K {us1_nosplit:7.1f} us ({fl / us1_nosplit / 1e6:6.0f} TF/s)", flush=True)


if __name__ == "__main__":
    main()
