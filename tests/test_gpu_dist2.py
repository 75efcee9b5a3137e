"""World size 2 on ONE GPU: two processes share cuda:0 and talk over gloo (RCCL refuses two ranks on one device), so
the real HIP kernels, the slab-overlapped asynchronous all-reduce and the relation partition run with a genuine peer.
The 8-GPU RCCL run itself belongs to the driver; this pins everything but the transport."""
import os
import socket

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, outdir, backend="gloo"):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (root, os.path.join(root, "torch-rgcn_amd")):
        sys.path.insert(0, p)
    import torch.distributed as dist
    from oracle import oracle
    from torch_rgcn.dist import shard_layer
    from torch_rgcn.layers import RelationalGraphConvolutionNC
    index = rank if backend == "nccl" else 0            # RCCL: one GPU per rank; gloo: both ranks share cuda:0
    torch.cuda.set_device(index)
    dev = torch.device("cuda", index)
    extra = {"device_id": dev} if backend == "nccl" else {}
    dist.init_process_group(backend, init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world, **extra)
    try:
        N, R0, E = 5000, 6, 90_000
        T = oracle.synthetic_triples(N, R0, E, seed=17)
        T[: E // 4, 0] = 5                                   # a hub: its tile is cut into several work units
        tp = torch.from_numpy(oracle.add_inverse_and_self(T, N, R0))
        res = {}
        for d_in, d_out in ((16, 16), (16, 4)):              # hidden-16 kernels and the padded output layer
            for sharded in (False, True):
                torch.manual_seed(0)
                layer = RelationalGraphConvolutionNC(triples=tp, num_nodes=N, num_relations=2 * R0 + 1,
                                                     in_features=d_in, out_features=d_out).to(dev)
                with torch.no_grad():
                    layer.bias.normal_()
                if sharded:
                    shard_layer(layer, dist.group.WORLD, keep="lpt")
                X = torch.randn(N, d_in, device=dev, requires_grad=True)
                out = layer(X)
                out.backward(torch.cos(out.detach()))
                wg = layer.weights.grad.clone()
                bg = layer.bias.grad.clone()
                if sharded:                                   # weight gradients stay with the owner: sum them for the check
                    dist.all_reduce(wg)
                res[(d_in, d_out, sharded)] = [t.detach().cpu().numpy() for t in (out, X.grad, wg, bg)]
                if sharded:
                    owned = layer._graph.owned_relations
                    assert 0 < len(owned) < 2 * R0 + 1
        if rank == 0:
            np.savez(os.path.join(outdir, "res.npz"), **{f"{k[0]}_{k[1]}_{int(k[2])}_{i}": a for k, v in res.items()
                                                        for i, a in enumerate(v)})
    finally:
        dist.destroy_process_group()


def test_two_ranks_one_gpu_sharded_layer_matches_unsharded(tmp_path):
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    z = np.load(os.path.join(str(tmp_path), "res.npz"))
    for d_in, d_out in ((16, 16), (16, 4)):
        for i, name in enumerate(("out", "dX", "dW", "db")):
            a, b = z[f"{d_in}_{d_out}_1_{i}"], z[f"{d_in}_{d_out}_0_{i}"]
            assert np.abs(a - b).max() <= 3e-5 * np.abs(b).max(), (d_in, d_out, name)


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs (the single-GPU boxes run the gloo variant above)")
def test_two_ranks_two_gpus_rccl_sharded_layer_matches_unsharded(tmp_path):
    """the same check over RCCL / xGMI when the box has two GPUs"""
    import torch.multiprocessing as mp
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    mp.spawn(_worker, args=(2, _free_port(), str(tmp_path), "nccl"), nprocs=2, join=True)
    z = np.load(os.path.join(str(tmp_path), "res.npz"))
    for d_in, d_out in ((16, 16), (16, 4)):
        for i, name in enumerate(("out", "dX", "dW", "db")):
            a, b = z[f"{d_in}_{d_out}_1_{i}"], z[f"{d_in}_{d_out}_0_{i}"]
            assert np.abs(a - b).max() <= 3e-5 * np.abs(b).max(), (d_in, d_out, name)


def test_bench_contract_with_two_ranks_on_one_gpu():
    """`torch.distributed.run --nproc-per-node 2 bench.py --gpus 2` exactly as the driver launches it, both ranks on
    cuda:0 over gloo, reduced workload: ONE JSON line, on rank 0, whole-job aggregate, weak scaling."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, RGCN_BENCH_ONE_DEVICE="1", RGCN_DIST_BACKEND="gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--nodes", "200000", "--edges", "1000000", "--rels", "20"]
    out = subprocess.run(cmd, env=env, cwd=root, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["steps"] == 3 and res["warmup"] == 1 and res["scaling"] == "weak"
    assert res["unit"] == "edges/s" and res["higher_is_better"] is True and res["dtype"] == "f32"
    assert abs(res["value"] - 2 * 1_000_000 / (res["ms_per_step"] * 1e-3)) < 1e-6 * res["value"]
    assert "cpu_baseline" not in res and res["roofline"]["bound"] == "hbm" and "relation-sharded x2" in res["config"]["sharding"]
