"""cProfile of the eager AIFB-shaped training step: where the host time goes when the kernels are short
(python tools/prof_step.py)."""
import cProfile, pstats, sys, os, io
ROOT=os.environ.get("GRAFT_REPO_ROOT","/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT,"torch-rgcn_amd"))
import torch
from torch_rgcn import _native
from torch_rgcn.models import NodeClassifier
N,R0,E=8285,45,29043
T=_native.synthetic_triples_host(N,R0,E,1)
m=NodeClassifier(triples=T,nnodes=N,nrel=R0,nhid=16,nlayers=2,nclass=4).cuda()
opt=torch.optim.Adam(m.parameters(),lr=0.01)
idx=torch.arange(176).cuda(); y=(torch.arange(176)%4).cuda()
def step():
    opt.zero_grad(set_to_none=True)
    loss=torch.nn.functional.cross_entropy(m()[idx],y)
    loss.backward(); opt.step()
for _ in range(5): step()
torch.cuda.synchronize()
pr=cProfile.Profile(); pr.enable()
for _ in range(200): step()
torch.cuda.synchronize(); pr.disable()
s=io.StringIO(); pstats.Stats(pr,stream=s).sort_stats("tottime").print_stats(28); print(s.getvalue()[:5000])
