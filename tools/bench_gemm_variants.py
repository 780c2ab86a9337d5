import sys, os, subprocess
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = '''
import sys; sys.path.insert(0, %r)
import torch
from neusky_amd import hip
dev="cuda:0"
def run(M,N,K,layout="nt",iters=30):
    if layout=="nt":
        A=torch.randn(M,K,device=dev); W=torch.randn(N,K,device=dev)/16; b=torch.randn(N,device=dev); C=torch.empty(M,N,device=dev)
        fn=lambda: hip.gemm(A,W,C,M,N,K,bias=b)
    elif layout=="nn":
        A=torch.randn(M,K,device=dev); W=torch.randn(K,N,device=dev)/16; C=torch.empty(M,N,device=dev)
        fn=lambda: hip.gemm(A,W,C,M,N,K,a_kcontig=True,b_kcontig=False)
    else:
        A=torch.randn(K,M,device=dev); W=torch.randn(K,N,device=dev); C=torch.zeros(M,N,device=dev)
        fn=lambda: hip.gemm(A,W,C,M,N,K,a_kcontig=False,b_kcontig=False,k_splits=128)
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    ms=e0.elapsed_time(e1)/iters
    return 2*M*N*K/ms/1e9
print("nt %%.1f  nn %%.1f  tn %%.1f  nt2560 %%.1f  sq4096 %%.1f" %% (run(262144,256,256), run(262144,256,256,"nn"), run(256,256,262144,"tn"), run(262144,2560,256,iters=5), run(4096,4096,4096,iters=10)))
''' % ROOT
for v in ["0","1","2","3"]:
    env=dict(os.environ, NSKY_GEMM_VARIANT=v)
    out=subprocess.run([sys.executable,"-c",code],env=env,capture_output=True,text=True)
    print("variant",v,out.stdout.strip(), out.stderr.strip()[-200:] if out.returncode else "")
