import torch, time, numpy as np, scipy.linalg as sla
dev=torch.device('cuda')
def tm(f, n=10):
    f(); torch.cuda.synchronize(); t=time.time()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.time()-t)/n*1e3
for m in (80,160,240):
    A=torch.randn(m,m,dtype=torch.float64,device=dev); A=A+A.T
    print('gpu eigh',m,'%.2f ms'%tm(lambda: torch.linalg.eigh(A)))
    for nt in (1,2,4,8):
        torch.set_num_threads(nt)
        def cpu():
            a=A.cpu(); w,v=torch.linalg.eigh(a); return w.to(dev,non_blocking=True),v.to(dev,non_blocking=True)
        print('  cpu torch threads',nt,'roundtrip %.2f ms'%tm(cpu))
    a=A.cpu().numpy()
    t=time.time()
    for _ in range(10): sla.eigh(a, driver='evd')
    print('  scipy evd only %.2f ms'%((time.time()-t)/10*1e3))
    t=time.time()
    for _ in range(10): sla.eigh(a, driver='evr', subset_by_index=[0,m//3-1])
    print('  scipy evr lowest third %.2f ms'%((time.time()-t)/10*1e3))
import os; print('cpus',os.cpu_count())
