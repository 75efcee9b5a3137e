import os, sys
sys.path.insert(0, "/root/repo/torch-rgcn_amd")
import numpy as np, torch
from torch_rgcn import _native
dev = torch.device("cuda:0")
N, d, B = 40_943, 200, 2
ag, G = torch.randn(N, B * d, device=dev), torch.randn(N, d, device=dev)
fl = 2.0 * N * B * d * d
def t(fn, iters=30):
    for _ in range(3): fn()
    torch.cuda.synchronize(); ts = []
    for _ in range(iters):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    return float(np.median(ts))
for bm in ("", "64", "128"):
    if bm: os.environ["RGCN_GEMM_BM"] = bm
    else: os.environ.pop("RGCN_GEMM_BM", None)
    for S in (8, 16, 24, 32, 48, 64, 96, 128):
        ms = t(lambda: _native.gemm(ag, G, trans_a=True, split_k=S))
        print(f"BM={bm or 'auto'} split_k={S}: {ms*1e3:.1f} us  {fl/ms/1e9:.1f} TF", flush=True)
