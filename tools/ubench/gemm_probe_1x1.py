import torch, time
dev='cuda:0'
def t(fn,n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter()-t0)/n*1e3
for (N,C,H,W,Co) in [(8,256,96,320,64),(8,16,96,320,64),(8,256,96,320,128)]:
    src=torch.randn(N,C,H,W,device=dev); w=torch.randn(Co,C,device=dev)
    a=t(lambda: torch.matmul(w, src.reshape(N,C,H*W)))
    b=t(lambda: torch.matmul(src.reshape(N,C,H*W).transpose(1,2), w.t()))
    c=t(lambda: torch.einsum('oc,nchw->nohw', w, src))
    d=t(lambda: torch.nn.functional.linear(src.permute(0,2,3,1), w))
    e=t(lambda: torch.matmul(w.unsqueeze(0).expand(N,Co,C).contiguous(), src.reshape(N,C,H*W)))
    print((N,C,H,W,Co), "matmul(w,src) %.3f | src^T w^T %.3f | einsum %.3f | linear(permute) %.3f | bmm contiguous %.3f ms"%(a,b,c,d,e))
