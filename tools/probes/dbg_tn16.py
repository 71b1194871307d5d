import os, sys
sys.path.insert(0, "/root/repo")
import torch
from lap_amd import hip
DEV="cuda"
def rnd(*s, seed=0, scale=1.0):
    g=torch.Generator().manual_seed(seed); return ((torch.rand(*s, generator=g)*2-1)*scale).to(DEV).bfloat16()
for M,N,K,pad in [(256,512,512,0),(2304,1280,512,0),(1024,768,1152,64)]:
    a=rnd(K,M+pad,seed=1)[:,:M]; b=rnd(K,N+pad,seed=2)[:,:N]
    outs=[]
    for tile in (12,14):
        out=torch.full((M,N+pad),3.0,device=DEV,dtype=torch.bfloat16)
        hip.gemm(a,b,out,M=M,N=N,K=K,lda=a.stride(0),ldb=b.stride(0),ldc=N+pad,a_kc=False,b_kc=False,tile=tile,ksplit=1)
        outs.append(out)
    ref=(a.float().t()@b.float())
    d=(outs[0].float()-outs[1].float())
    print(M,N,K,"equal",torch.equal(outs[0],outs[1]),"maxdiff",d.abs().max().item(),"hip err",((outs[0][:,:N].float()-ref).norm()/ref.norm()).item(),"asm err",((outs[1][:,:N].float()-ref).norm()/ref.norm()).item(), "pad ok", bool((outs[1][:,N:]==3.0).all()))
    bad=(d[:,:N]!=0).nonzero()
    print("  bad count",len(bad),"first",bad[:5].tolist(), "rows", sorted(set((bad[:,0]//16).tolist()))[:10], "cols", sorted(set((bad[:,1]//16).tolist()))[:10])
    # transposed?
    if M==N: print("  transposed match", torch.equal(outs[1][:,:N], outs[0][:,:N].t()))
