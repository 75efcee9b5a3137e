"""More than 2^24 nodes (the packed 8-byte slot holds a 24-bit source id): the unpacked slot arrays against the oracle
at N = 17 M (python tools/big_n_check.py; ~20 s of oracle time); `python tools/big_n_check.py 30000000 40 5000000 16` = 2.43e9
(relation, node) cells, past 2^31 (64-bit cell indices in the device-side graph build)."""
import sys, os, time
ROOT=os.environ.get("GRAFT_REPO_ROOT","/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT,"torch-rgcn_amd"))
import numpy as np, torch
from oracle import oracle
from torch_rgcn.layers import RelationalGraphConvolutionNC
N,R0,E,d=(int(a) for a in (sys.argv[1:5] if len(sys.argv) > 4 else ("17000000","10","5000000","16")))
T=oracle.synthetic_triples(N,R0,E,seed=1); tp=oracle.add_inverse_and_self(T,N,R0); R=2*R0+1
layer=RelationalGraphConvolutionNC(triples=torch.from_numpy(tp),num_nodes=N,num_relations=R,in_features=d,out_features=d).cuda()
rng=np.random.default_rng(0)
X=torch.from_numpy(rng.standard_normal((N,d)).astype(np.float32)).cuda().requires_grad_(True)
t=time.time(); out=layer(X); torch.cuda.synchronize(); print("first forward incl. build", round(time.time()-t,2), "packed:", layer._graph.fwd_plan(16).pack is not None, flush=True)
g=rng.standard_normal((N,d)).astype(np.float32)
out.backward(torch.from_numpy(g).cuda()); torch.cuda.synchronize()
ref=oracle.nc_layer(tp,N,R,X.detach().cpu().numpy(),{"weights":layer.weights.detach().cpu().numpy()},"none",layer.bias.detach().cpu().numpy(),False,g)
def rel(a,b):
    a=a.detach().cpu().numpy().astype(np.float64); return float(np.abs(a-b).max()/np.abs(b).max())
print("rel errs", rel(out,ref["out"]), rel(X.grad,ref["dX"]), rel(layer.weights.grad,ref["grads"]["weights"]), rel(layer.bias.grad, ref["db"]))
