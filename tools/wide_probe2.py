import os, sys, time, json
ROOT=os.environ.get("GRAFT_REPO_ROOT","/root/repo")
sys.path.insert(0,os.path.join(ROOT,"torch-rgcn_amd"))
import numpy as np, torch
from torch_rgcn import _native
from torch_rgcn.layers import RelationalGraphConvolutionLP, RelationalGraphConvolutionNC
dev=torch.device("cuda")
def timed(fn,iters=10,warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); ts=[]
    for _ in range(iters):
        t0=time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter()-t0)
    return 1e3*float(np.median(ts))
for tag,N,R0,E,d in (("WN18-shaped undecomposed d=200",40943,18,15000,200),("FB15k-237-shaped undecomposed d=100",14541,237,30000,100)):
    layer=RelationalGraphConvolutionLP(num_nodes=N,num_relations=2*R0+1,in_features=d,out_features=d,edge_dropout={"general":0.5,"self_loop":0.2,"self_loop_type":"schlichtkrull-dropout"},decomposition=None,w_init="glorot-normal",b_init="zeros").to(dev).eval()
    T=torch.from_numpy(_native.synthetic_triples_host(N,R0,E,3)).to(dev)
    X=torch.randn(N,d,device=dev,requires_grad=True)
    def step():
        X.grad=None
        for p in layer.parameters(): p.grad=None
        layer(T,X).pow(2).mean().backward()
    res={}
    for mode in ("gemm","blocks"):
        os.environ["RGCN_WIDE"]=mode
        res[mode]=round(timed(step),3)
        _native.profile_start(); step(); prof=_native.profile_stop()
        res[mode+"_kernels"]={k:round(float(np.sum(v)),3) for k,v in prof.items()}
    M=3*E+N
    res["flops_fwd"]=2.0*M*d*d
    print(json.dumps({"workload":tag,"N":N,"R":2*R0+1,"messages":M,**res}))
