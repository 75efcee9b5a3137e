"""Which ATen / autograd op launched which kernel in one WN18-shaped LP training step (torch.profiler, input shapes recorded): how the two
float -> bool -> uint8 casts of the dropout mask (8 us each) and the embeddings' ReLU pair were found.  python tools/optrace.py"""
import sys, torch
sys.path.insert(0, "torch-rgcn_amd"); sys.path.insert(0, "tools")
from torch_rgcn import _native
from torch_rgcn.layers import RelationalGraphConvolutionLP, DistMult
from torch_rgcn.functional import bce_with_logits, unit_gradient
DEV = torch.device("cuda:0")
N, R0, d, E, Tn = 40_943, 18, 200, 15_000, 330_000
ed = {"general": 0.5, "self_loop": 0.2, "self_loop_type": "schlichtkrull-dropout"}
layer = RelationalGraphConvolutionLP(num_nodes=N, num_relations=2 * R0 + 1, in_features=d, out_features=d, edge_dropout=ed,
                                     decomposition={"type": "basis", "num_bases": 2}, w_init="glorot-normal", b_init="zeros").to(DEV)
dm = DistMult(R0, d, N, R0).to(DEV)
emb = torch.randn(N, d, device=DEV, requires_grad=True)
graph = torch.from_numpy(_native.synthetic_triples_host(N, R0, E, 3)).to(DEV)
batch = torch.from_numpy(_native.synthetic_triples_host(N, R0, Tn, 4)).to(DEV)
y = torch.rand(Tn, device=DEV).round()
unit = unit_gradient(DEV)
def step():
    for p in [emb] + list(layer.parameters()) + list(dm.parameters()):
        p.grad = None
    x = layer(graph, torch.relu(emb))
    bce_with_logits(dm(batch, x), y).backward(gradient=unit)
for _ in range(3): step()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    step(); torch.cuda.synchronize()
evs = [e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CPU and e.cuda_time_total > 0 or True]
print(prof.key_averages(group_by_input_shape=True).table(sort_by="cuda_time_total", row_limit=40, max_name_column_width=50, max_shapes_column_width=60))
