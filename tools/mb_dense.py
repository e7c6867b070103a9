import torch, time
dev=torch.device('cuda')
def tm(f, n=10):
    f(); torch.cuda.synchronize(); t=time.time()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.time()-t)/n*1e3
for lib in ['default','magma']:
    try:
        torch.backends.cuda.preferred_linalg_library(lib)
    except Exception as e: print('lib',lib,e); continue
    for m in (80,160,240):
        A=torch.randn(m,m,dtype=torch.float64,device=dev); A=A+A.T
        print(lib,'eigh',m,'%.2f ms'%tm(lambda: torch.linalg.eigh(A)))
    B=torch.randn(160,80,dtype=torch.float64,device=dev)
    print(lib,'qr 160x80 %.2f ms'%tm(lambda: torch.linalg.qr(B)))
    C=torch.randn(80,80,dtype=torch.float64,device=dev); C=C@C.T+80*torch.eye(80,dtype=torch.float64,device=dev)
    print(lib,'chol 80 %.2f ms'%tm(lambda: torch.linalg.cholesky(C)))
torch.backends.cuda.preferred_linalg_library('default')
for m in (80,240):
    A=torch.randn(m,m,dtype=torch.float64,device=dev); A=A+A.T
    def cpu():
        a=A.cpu(); w,v=torch.linalg.eigh(a); return w.to(dev),v.to(dev)
    print('cpu roundtrip eigh',m,'%.2f ms'%tm(cpu))
    a=A.cpu()
    t=time.time(); 
    for _ in range(10): torch.linalg.eigh(a)
    print('cpu only eigh',m,'%.2f ms'%((time.time()-t)/10*1e3), 'threads',torch.get_num_threads())
n=446631
X=torch.randn(n,240,device=dev); Y=torch.randn(n,240,device=dev)
print('rocblas f32 X^T Y 240x240 %.2f ms'%tm(lambda: X.T@Y))
X64=X.double(); Y64=Y.double()
print('rocblas f64 X^T Y 240x240 %.2f ms'%tm(lambda: X64.T@Y64))
Cm=torch.randn(240,80,device=dev)
print('rocblas f32 X@C (240->80) %.2f ms'%tm(lambda: X@Cm))
Cm2=torch.randn(240,160,device=dev)
print('rocblas f32 X@C (240->160) %.2f ms'%tm(lambda: X@Cm2))
print('copy n x240 %.2f ms'%tm(lambda: X.clone()))
