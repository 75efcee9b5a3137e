"""rgcn_colsum_f32 (bias gradients) against torch.sum(0) at the shapes of the configs: narrow rows (hidden 16 / 32), wide rows
(d = 200, 500), the DistMult-sized 330,000 x 200."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "torch-rgcn_amd"))
from torch_rgcn import _native
dev = torch.device("cuda")
def timed(fn, iters=50, warm=5):
    for _ in range(warm): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3
for n, d in ((40943, 200), (14545, 500), (1000000, 16), (1666764, 32), (40943, 64), (40943, 128), (330000, 200)):
    G = torch.randn(n, d, device=dev)
    ref = G.sum(0)
    out = _native.colsum(G)
    print(n, d, "colsum us", round(timed(lambda: _native.colsum(G)), 1), "torch.sum us", round(timed(lambda: G.sum(0)), 1), "err", float((out - ref).abs().max() / ref.abs().max()))
