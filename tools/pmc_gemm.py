import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from neusky_amd import hip
dev="cuda:0"
M,N,K=262144,256,256
A=torch.randn(M,K,device=dev); W=torch.randn(N,K,device=dev)/16; b=torch.randn(N,device=dev); C=torch.empty(M,N,device=dev)
for _ in range(5): hip.gemm(A,W,C,M,N,K,bias=b)
torch.cuda.synchronize()
