"""Undecomposed weights at widths above 16: per-message kernels vs transform-then-aggregate through the R x N x d message
table, fwd+bwd, three graph shapes (python tools/wide_probe.py)."""
import sys, os, time
ROOT=os.environ.get("GRAFT_REPO_ROOT","/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT,"torch-rgcn_amd"))
import torch
from torch_rgcn import _native, functional as F_
from torch_rgcn.layers import RelationalGraphConvolutionNC
def timed(fn, it=5):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(it): fn()
    torch.cuda.synchronize(); return (time.perf_counter()-t)/it*1e3
for (N,R0,E,d) in ((40943,18,141442,200),(14541,237,272115,100),(8285,45,29043,64)):
    T=_native.synthetic_triples_host(N,R0,E,1)
    tp=torch.from_numpy(_native.add_inverse_and_self_host(T,N,R0))
    layer=RelationalGraphConvolutionNC(triples=tp,num_nodes=N,num_relations=2*R0+1,in_features=d,out_features=d).cuda()
    X=torch.randn(N,d,device="cuda",requires_grad=True)
    def step():
        X.grad=None; layer.zero_grad()
        layer(X).pow(2).mean().backward()
    a=timed(step)
    graph=layer._graph
    def step2():
        X.grad=None; layer.zero_grad()
        table=torch.matmul(X, layer.weights)      # [R,N,d]
        F_.featureless_mp(table, layer.bias, graph).pow(2).mean().backward()
    b=timed(step2)
    print(f"N={N} R={2*R0+1} M={tp.shape[0]} d={d}: per-message generic {a:.2f} ms   transform-then-aggregate {b:.2f} ms", flush=True)
