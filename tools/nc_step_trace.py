"""Driver for kernel traces of the secondary config steps (run under rocprofv3 --kernel-trace --stats):
  python tools/nc_step_trace.py am | mutag | aifb      -> 50 eager steps of the config line's step"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tools"))
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "torch-rgcn_amd"))
import config_bench as cb  # noqa: E402
from torch_rgcn import _native  # noqa: E402
from torch_rgcn.layers import RelationalGraphConvolutionNC  # noqa: E402
from torch_rgcn.models import NodeClassifier  # noqa: E402

DEV = torch.device("cuda")
which = sys.argv[1] if len(sys.argv) > 1 else "am"
if which == "am":
    N, R0, E, d = 1_666_764, 133, 5_988_321, 16
    T = _native.synthetic_triples_host(N, R0, E, 2)
    tp = torch.from_numpy(_native.add_inverse_and_self_host(T, N, R0))
    kw = dict(triples=tp, num_nodes=N, num_relations=2 * R0 + 1, in_features=d, out_features=d, decomposition={"type": "block", "num_blocks": 4})
    l1 = RelationalGraphConvolutionNC(vertical_stacking=False, **kw).to(DEV)
    l2 = RelationalGraphConvolutionNC(vertical_stacking=True, **kw).to(DEV)
    X = torch.randn(N, d, device=DEV, requires_grad=True)

    def step():
        for p in [X] + list(l1.parameters()) + list(l2.parameters()):
            p.grad = None
        l2(l1.forward_activated(X, "relu", private=True)).pow(2).mean().backward()
else:
    N, R0, E, nhid, ncls, decomp, lab = {"aifb": (8285, 45, 29_043, 16, 4, None, 140),
                                         "mutag": (23_644, 23, 74_227, 16, 2, {"type": "basis", "num_bases": 30}, 272)}[which]
    T = _native.synthetic_triples_host(N, R0, E, 1)
    model = NodeClassifier(triples=T, nnodes=N, nrel=R0, nhid=nhid, nclass=ncls, decomposition=decomp).to(DEV)
    idx = torch.arange(lab, device=DEV)
    y = torch.randint(0, ncls, (lab,), device=DEV)
    opt = torch.optim.Adam(model.parameters(), lr=0.01, fused=True)

    from torch_rgcn.functional import MaskedCrossEntropy, unit_gradient
    head = MaskedCrossEntropy(idx, y, N)
    unit = unit_gradient(DEV)

    def step():
        opt.zero_grad(set_to_none=True)
        head(model()).backward(gradient=unit)      # as experiments/classify_nodes.py does
        opt.step()
for _ in range(5):
    step()
torch.cuda.synchronize()
for _ in range(50):
    step()
torch.cuda.synchronize()
print("done", which)
