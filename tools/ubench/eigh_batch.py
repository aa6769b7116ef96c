import torch, time
dev="cuda"
torch.manual_seed(0)
n=208
A=torch.randn(32,n,n,dtype=torch.float64,device=dev); S=A@A.transpose(-1,-2)/n+torch.eye(n,dtype=torch.float64,device=dev)
for f,name in ((lambda: [torch.linalg.eigh(S[i]) for i in range(32)],"32 single"),(lambda: torch.linalg.eigh(S),"batched 32"),(lambda: torch.linalg.cholesky(S),"batched cholesky"), (lambda: [torch.linalg.eigh(S[i].cpu()) for i in range(32)],"32 single on CPU")):
    f(); torch.cuda.synchronize()
    t0=time.perf_counter()
    for _ in range(3): f()
    torch.cuda.synchronize()
    print(name, "%.2f ms"%((time.perf_counter()-t0)/3*1e3))
