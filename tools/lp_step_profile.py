"""Host-side profile of the eager WN18-shaped LP step (where do the ~0.5 ms between the eager step and its hipGraph replay go?):
cProfile over 100 steps + the device -> host synchronisations torch reports (set_sync_debug_mode("warn"))."""
import cProfile
import io
import os
import pstats
import sys
import warnings

import torch

sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "torch-rgcn_amd"))
from torch_rgcn import _native  # noqa: E402
from torch_rgcn.layers import DistMult, RelationalGraphConvolutionLP  # noqa: E402

DEV = torch.device("cuda")
N, R0, d, E, Tn = 40_943, 18, 200, 15_000, 330_000
ed = {"general": 0.5, "self_loop": 0.2, "self_loop_type": "schlichtkrull-dropout"}
layer = RelationalGraphConvolutionLP(num_nodes=N, num_relations=2 * R0 + 1, in_features=d, out_features=d, edge_dropout=ed,
                                     decomposition={"type": "basis", "num_bases": 2}, w_init="glorot-normal", b_init="zeros").to(DEV)
dm = DistMult(R0, d, N, R0).to(DEV)
emb = torch.randn(N, d, device=DEV, requires_grad=True)
graph = torch.from_numpy(_native.synthetic_triples_host(N, R0, E, 3)).to(DEV)
batch = torch.from_numpy(_native.synthetic_triples_host(N, R0, Tn, 4)).to(DEV)
y = torch.rand(Tn, device=DEV).round()


def step():
    for p in [emb] + list(layer.parameters()) + list(dm.parameters()):
        p.grad = None
    x = layer(graph, torch.relu(emb))
    torch.nn.functional.binary_cross_entropy_with_logits(dm(batch, x), y).backward()


for _ in range(5):
    step()
torch.cuda.synchronize()
with warnings.catch_warnings(record=True) as w:
    warnings.simplefilter("always")
    torch.cuda.set_sync_debug_mode("warn")
    step()
    torch.cuda.set_sync_debug_mode("default")
print("synchronising calls in one step:", len(w))
for m in w[:12]:
    print("   ", str(m.message)[:110], "@", os.path.basename(m.filename), m.lineno)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(100):
    step()
torch.cuda.synchronize()
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(22)
print(s.getvalue()[:6000])
