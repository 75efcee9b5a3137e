import os, sys, time, json
ROOT=os.environ.get("GRAFT_REPO_ROOT","/root/repo")
sys.path.insert(0,os.path.join(ROOT,"torch-rgcn_amd"))
import numpy as np, torch
from torch_rgcn import _native
from torch_rgcn.layers import RelationalGraphConvolutionLP
dev=torch.device("cuda")
def timed(fn,iters=5,warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); ts=[]
    for _ in range(iters):
        t0=time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter()-t0)
    return 1e3*float(np.median(ts))
for tag,N,R0,E,d,nb in (("FB15k-237-shaped block 100 x (5x5), d=500",14545,237,30000,500,100),("FB-toy-shaped",280,112,300,500,100)):
    res={}
    for mode in ("1","0"):
        os.environ["RGCN_BLOCK_TABLE"]=mode
        layer=RelationalGraphConvolutionLP(num_nodes=N,num_relations=2*R0+1,in_features=d,out_features=d,edge_dropout={"general":0.5,"self_loop":0.2,"self_loop_type":"schlichtkrull-dropout"},decomposition={"type":"block","num_blocks":nb},b_init="zeros").to(dev).eval()
        T=torch.from_numpy(_native.synthetic_triples_host(N,R0,E,3)).to(dev)
        X=torch.randn(N,d,device=dev,requires_grad=True)
        def step():
            X.grad=None
            for p in layer.parameters(): p.grad=None
            layer(T,X).pow(2).mean().backward()
        try:
            res["table" if mode=="1" else "dense_gemm"]=round(timed(step),3)
            torch.cuda.reset_peak_memory_stats(); step(); torch.cuda.synchronize()
            res[("table" if mode=="1" else "dense_gemm")+"_peak_GB"]=round(torch.cuda.max_memory_allocated()/1e9,2)
        except Exception as exc:
            res["table" if mode=="1" else "dense_gemm"]=f"{type(exc).__name__}: {exc}"[:120]
        del layer; torch.cuda.empty_cache()
    print(json.dumps({"workload":tag,"N":N,"R":2*R0+1,"messages":3*E+N,**res}))
