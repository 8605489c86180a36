import os, sys, torch
sys.path.insert(0, os.getcwd())
from jepa_amd.hip import ops
A = torch.randn(37632, 1024, device="cuda").bfloat16(); W = torch.randn(1024, 1024, device="cuda").bfloat16()
for fl in (0x100, 0xC0):
    for _ in range(3): ops.gemm_nt(A, W, flags=fl)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20): ops.gemm_nt(A, W, flags=fl)
    e.record(); torch.cuda.synchronize()
    print(hex(fl), s.elapsed_time(e) / 20 * 1e3, "us")
