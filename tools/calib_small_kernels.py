import torch, numpy as np
dev="cuda:0"
def t(fn,n=30):
    fn(); torch.cuda.synchronize(); ts=[]
    for i in range(n):
        a,b=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b)*1e3)
    return np.median(ts)
big=torch.randn(64*1024*1024//8,device=dev,dtype=torch.float64)   # 64 MB to evict caches
for T in (1000,10000,100000,1000000):
    x=torch.randn(T,9,device=dev,dtype=torch.float64); y=torch.empty_like(x)
    print("T=%7d  y=x*2 (read %5.1f MB, write same): %.1f us warm | after cache flush %.1f us"%(T,T*72/1e6,t(lambda: torch.mul(x,2.0,out=y)), t(lambda: (big.add_(1.0), torch.mul(x,2.0,out=y))[1]) - t(lambda: big.add_(1.0))))
print("empty event pair: %.1f us"%t(lambda: None))
