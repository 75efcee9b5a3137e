"""Where the time of rgcn_gemm_f32 goes at WN18-like shapes: workgroups per CU (128 x 128 tiles over 256 CUs), ragged N, K.
One JSON line per shape; us_per_slab_round = time / (K / 16) / ceil(tiles / 256)."""
import os, sys, json, torch
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "torch-rgcn_amd"))
from torch_rgcn import _native
dev = torch.device("cuda")
def timed(fn, iters=30, warm=5):
    for _ in range(warm): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters
for M, N, K in ((40943,200,400),(40943,256,400),(40943,128,400),(81886,200,400),(163772,200,400),(40943,200,1600),(32768,256,400),(65536,256,400),(16384,256,400)):
    A = torch.randn(M, K, device=dev); B = torch.randn(K, N, device=dev)
    ms = timed(lambda: _native.gemm(A, B))
    tiles = -(-M//128) * -(-N//128)
    print(json.dumps({"M":M,"N":N,"K":K,"ms":round(ms,4),"TF":round(2.0*M*N*K/ms/1e9,1),"tiles":tiles,"tiles_per_CU":round(tiles/256,2), "us_per_slab_round": round(ms*1e3/(K/16)/max(1,-(-tiles//256)),3)}))
