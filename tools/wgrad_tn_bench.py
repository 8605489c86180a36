import os, sys, torch
sys.path.insert(0, os.getcwd())
from jepa_amd.hip import ops
dev="cuda"
g=torch.Generator(device=dev).manual_seed(0)
for tag,T,N1,N2 in [("qkv",11392,3072,1024),("proj",11392,1024,1024),("fc1",11392,4096,1024),("fc2",11392,1024,4096)]:
    dY=torch.randn(T,N1,device=dev,generator=g).to(torch.bfloat16); X=torch.randn(T,N2,device=dev,generator=g).to(torch.bfloat16)
    out=torch.empty(N1,N2,device=dev)
    dYt,Xt=ops.transpose(dY),ops.transpose(X)
    def t(fn):
        for _ in range(3): fn()
        s,e=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(20): fn()
        e.record(); torch.cuda.synchronize()
        return s.elapsed_time(e)*1e3/20
    a=t(lambda: ops.gemm_wgrad_tn(dY,X,out))
    b=t(lambda: ops.gemm_wgrad(dYt,Xt,out))
    c=t(lambda: (ops.transpose(dY),ops.transpose(X)))
    fl=2.0*T*N1*N2
    print(f"{tag}: TN {a:7.1f} us {fl/a/1e6:7.1f} TF/s | NT {b:7.1f} us {fl/b/1e6:7.1f} TF/s | transposes {c:6.1f} us")
