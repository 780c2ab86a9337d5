import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from neusky_amd import hip
dev="cuda:0"
def run(M,N,K,layout,prec,iters=20,**kw):
    if layout=="nt":
        A=torch.randn(M,K,device=dev); W=torch.randn(N,K,device=dev)/16; C=torch.empty(M,N,device=dev)
        fn=lambda: hip.gemm(A,W,C,M,N,K,precision=prec,**kw)
    elif layout=="nn":
        A=torch.randn(M,K,device=dev); W=torch.randn(K,N,device=dev)/16; C=torch.empty(M,N,device=dev)
        fn=lambda: hip.gemm(A,W,C,M,N,K,a_kcontig=True,b_kcontig=False,precision=prec,**kw)
    else:
        A=torch.randn(K,M,device=dev); W=torch.randn(K,N,device=dev); C=torch.zeros(M,N,device=dev)
        fn=lambda: hip.gemm(A,W,C,M,N,K,a_kcontig=False,b_kcontig=False,k_splits=128,precision=prec,**kw)
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    ms=e0.elapsed_time(e1)/iters
    return ms, 2*M*N*K/ms/1e9
for prec in (0,2,3):
    r=[run(262144,256,256,"nt",prec), run(262144,256,256,"nn",prec), run(256,256,262144,"tn",prec), run(262144,256,2560,"nn",prec,iters=5), run(2560,256,262144,"tn",prec,iters=5), run(4096,4096,4096,"nt",prec,iters=10)]
    print("prec",prec," | ".join(f"{ms:.3f}ms {tf:.0f}TF" for ms,tf in r))
