"""End-to-end: the experiment counterparts train on dataset-shaped synthetic graphs through the HIP layers."""
import os

import numpy as np
import sys

import pytest
import yaml

from conftest import PKG

pytestmark = pytest.mark.gpu


def cfg(name):
    return yaml.safe_load(open(os.path.join(PKG, "configs", "rgcn", name)))


def test_classify_nodes_aifb_shaped_learns():
    sys.path.insert(0, os.path.join(PKG, "experiments"))
    import classify_nodes
    hist = classify_nodes.run(cfg("nc-AIFB.yaml"), epochs=25, quiet=True, synthetic=True)
    assert hist[-1][0] < 0.6 * hist[0][0]          # loss goes down
    assert hist[-1][1] > 0.9                         # and the (featureless, 12 M parameter) model fits the train labels


def test_classify_nodes_hipgraph_replay_matches_eager():
    """the captured training step is the same computation, from the same state (the warm-up steps of the capture are undone): the
    loss trajectory of the DEFAULT path (capture unless it fails), of hipgraph=True and of the eager loop agree epoch by epoch"""
    sys.path.insert(0, os.path.join(PKG, "experiments"))
    import classify_nodes
    import torch
    hist = {}
    for mode in (False, True, None):
        torch.manual_seed(0)
        hist[mode] = classify_nodes.run(cfg("nc-MUTAG.yaml"), epochs=8, quiet=True, hipgraph=mode, synthetic=True)
    assert len(hist[True]) == 8 and hist[True][-1][0] < hist[True][0][0]
    # (fp32 atomics in the weight gradients make the trajectories drift apart in the last digits, amplified by Adam over the epochs)
    for mode in (True, None):
        for k in range(8):
            assert abs(hist[mode][k][0] - hist[False][k][0]) < 2e-2 * abs(hist[False][k][0]), (mode, k, hist[mode][k][0], hist[False][k][0])
        assert abs(hist[mode][0][0] - hist[False][0][0]) < 1e-4 * abs(hist[False][0][0]), "epoch 1 starts from the same state"


@pytest.mark.parametrize("tile_mode", ["nodes", "ranges"])
def test_classify_nodes_capture_with_the_in_place_tile_kernels(monkeypatch, tile_mode):
    """the same with the featureless basis layer on the in-place tile kernels (rgcn_fbasis_tile.hip: the route of tables beyond the caches --
    AM as shipped --, forced here on the MUTAG-shaped model), both forms: the captured step replays them (LDS above 64 KB, persistent
    workgroups, the plan's largest source degree read before the capture) and follows the eager trajectory"""
    sys.path.insert(0, os.path.join(PKG, "experiments"))
    import classify_nodes
    import torch
    from torch_rgcn import _native, routes
    routes.patch(monkeypatch, "fbasis_inplace_mb", "0")
    routes.patch(monkeypatch, "fbasis_tile", tile_mode)
    hist = {}
    for mode in (False, None):
        torch.manual_seed(0)
        if mode is False:                       # (the per-kernel timers record events: not inside a capture)
            _native.profile_start()
        hist[mode] = classify_nodes.run(cfg("nc-MUTAG.yaml"), epochs=6, quiet=True, hipgraph=mode, synthetic=True)
        if mode is False:
            prof = _native.profile_stop()
            assert "fbasis_tile_fwd" in prof and "fbasis_tile_bwd" in prof, sorted(prof)
    assert hist[None][-1][0] < hist[None][0][0]
    for k in range(6):
        assert abs(hist[None][k][0] - hist[False][k][0]) < 2e-2 * abs(hist[False][k][0]), (k, hist[None][k][0], hist[False][k][0])


def test_experiments_default_to_the_captured_step(monkeypatch):
    """the default run of both experiments replays a captured hipGraph (route capture=0 / --eager: the reference's loop)"""
    sys.path.insert(0, os.path.join(PKG, "experiments"))
    import classify_nodes
    import predict_links
    import torch
    replays = []
    orig = torch.cuda.CUDAGraph.replay
    monkeypatch.setattr(torch.cuda.CUDAGraph, "replay", lambda self: (replays.append(1), orig(self))[1])
    classify_nodes.run(cfg("nc-AIFB.yaml"), epochs=3, quiet=True, synthetic=True)
    assert len(replays) == 6, "3 training replays + 3 evaluation replays"
    del replays[:]
    c = cfg("lp-WN18.yaml")
    c["dataset"]["name"] = "fb-toy"
    c["encoder"].update(node_embedding=32, hidden1_size=32)
    c["training"].update(graph_batch_size=500)
    c["evaluation"].update(check_every=1000, batch_size=32, verbose=False)
    predict_links.run(c, epochs=4, quiet=True, max_test=10, synthetic=True)
    assert len(replays) == 4
    del replays[:]
    from torch_rgcn import routes
    with routes.override(capture="0"):
        classify_nodes.run(cfg("nc-AIFB.yaml"), epochs=2, quiet=True, synthetic=True)
    assert not replays


def test_classify_nodes_mutag_shaped_basis():
    sys.path.insert(0, os.path.join(PKG, "experiments"))
    import classify_nodes
    hist = classify_nodes.run(cfg("nc-MUTAG.yaml"), epochs=15, quiet=True, synthetic=True)
    assert hist[-1][0] < hist[0][0]


def test_classify_nodes_e_rgcn_config_runs():
    sys.path.insert(0, os.path.join(PKG, "experiments"))
    import classify_nodes
    c = yaml.safe_load(open(os.path.join(PKG, "configs", "e-rgcn", "nc-AIFB.yaml")))
    hist = classify_nodes.run(c, epochs=10, quiet=True, synthetic=True)
    assert hist[-1][0] < hist[0][0]


def test_predict_links_small_graph_trains_and_ranks():
    sys.path.insert(0, os.path.join(PKG, "experiments"))
    import predict_links
    c = cfg("lp-WN18.yaml")                              # reference schema: edge-neighbourhood sampling, 10 negatives, basis 2
    c["dataset"]["name"] = "fb-toy"
    c["encoder"].update(node_embedding=32, hidden1_size=32)
    c["training"].update(graph_batch_size=2000)
    c["evaluation"].update(check_every=20, batch_size=32, verbose=False)
    hist, metrics = predict_links.run(c, epochs=30, quiet=True, max_test=100, synthetic=True)
    assert hist[-1] < hist[0]
    assert 0.0 < metrics["mrr"] <= 1.0 and metrics["hits@10"] >= metrics["hits@1"]


def test_predict_links_block_config_pads_nodes_and_c_rgcn_runs():
    sys.path.insert(0, os.path.join(PKG, "experiments"))
    import predict_links
    c = cfg("lp-FB-toy.yaml")                            # block decomposition: 280 nodes padded to a multiple of 500 / 100
    c["training"].update(graph_batch_size=300)
    hist, metrics = predict_links.run(c, epochs=3, quiet=True, max_test=20, synthetic=True)
    assert len(hist) == 3 and 0.0 < metrics["mrr"] <= 1.0
    c = yaml.safe_load(open(os.path.join(PKG, "configs", "c-rgcn", "lp-FB-toy.yaml")))   # no graph_batch_size: whole graph
    hist, metrics = predict_links.run(c, epochs=3, quiet=True, max_test=20, synthetic=True)
    assert len(hist) == 3 and 0.0 < metrics["mrr"] <= 1.0


def test_predict_links_training_step_as_hipgraph_learns():
    """the WN18-config training step (basis 2, per-step graph build, DistMult, schlichtkrull-l2 penalty,
    Adam) captured once in a hipGraph and replayed on freshly sampled batches"""
    sys.path.insert(0, os.path.join(PKG, "experiments"))
    import predict_links
    c = cfg("lp-WN18.yaml")
    c["dataset"]["name"] = "fb-toy"
    c["encoder"].update(node_embedding=64, hidden1_size=64)
    c["training"].update(graph_batch_size=2000)
    c["evaluation"].update(check_every=10, batch_size=32, verbose=False)    # ranking evaluations (eager kernels) between replays
    hist, metrics = predict_links.run(c, epochs=40, quiet=True, max_test=50, synthetic=True, hipgraph=True)
    assert len(hist) == 40 and all(np.isfinite(hist)) and np.mean(hist[-5:]) < np.mean(hist[:5])
    assert 0.0 < metrics["mrr"] <= 1.0


def _replay_safe_in_a_fresh_process(prelude, env_value):
    """REPLAY_SAFE as a fresh interpreter sees it: `prelude` runs before `import torch_rgcn`; env_value: the variable at exec time (None: unset)"""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k != "DEBUG_CLR_GRAPH_PACKET_CAPTURE"}
    if env_value is not None:
        env["DEBUG_CLR_GRAPH_PACKET_CAPTURE"] = env_value
    code = f"import sys; sys.path.insert(0, {PKG!r}); import torch\n{prelude}\nimport torch_rgcn; print('SAFE', torch_rgcn.REPLAY_SAFE)"
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-1500:]
    return out.stdout.strip().splitlines()[-1] == "SAFE True"


def test_replay_safe_is_not_inferred_from_a_variable_set_too_late():
    """captured steps are the default only when the runtime's graph packet capture is REALLY off.  The runtime reads
    DEBUG_CLR_GRAPH_PACKET_CAPTURE once, when it starts -- torch.cuda.is_available() / device_count() start it without setting torch's
    lazy-init flag, and a setdefault after that does nothing: REPLAY_SAFE must then be False (the experiments fall back to the eager loop)
    unless the variable was 0 in the environment the process started with."""
    assert _replay_safe_in_a_fresh_process("", None) is True                                   # import first: the module switches it off in time
    assert _replay_safe_in_a_fresh_process("torch.cuda.is_available()", None) is False         # runtime already up: too late
    assert _replay_safe_in_a_fresh_process("torch.zeros(1, device='cuda')", None) is False
    assert _replay_safe_in_a_fresh_process("torch.cuda.is_available()", "0") is True           # off from the start: safe whenever we look
    assert _replay_safe_in_a_fresh_process("", "1") is False                                   # the user asked for the feature
    late = "torch.cuda.is_available(); import os; os.environ.setdefault('DEBUG_CLR_GRAPH_PACKET_CAPTURE', '0')"
    assert _replay_safe_in_a_fresh_process(late, None) is False                                # somebody else's setdefault after the start


def test_classify_nodes_eager_fallback_starts_from_the_initial_state(monkeypatch):
    """when the capture fails the eager loop must start epoch 1 from the initial parameters and a fresh optimiser, not after
    the warm-up's three steps: same first-epoch loss and trajectory as hipgraph=False"""
    sys.path.insert(0, os.path.join(PKG, "experiments"))
    import classify_nodes
    import torch
    torch.manual_seed(0)
    ref = classify_nodes.run(cfg("nc-MUTAG.yaml"), epochs=4, quiet=True, hipgraph=False, synthetic=True)

    class Broken:
        def __init__(self, *a, **k):
            raise RuntimeError("capture is broken in this test")
    monkeypatch.setattr(torch.cuda, "CUDAGraph", Broken)
    torch.manual_seed(0)
    with pytest.warns(UserWarning, match="capture of the training step failed"):
        got = classify_nodes.run(cfg("nc-MUTAG.yaml"), epochs=4, quiet=True, hipgraph=None, synthetic=True)
    assert abs(got[0][0] - ref[0][0]) < 1e-4 * abs(ref[0][0]), (got[0][0], ref[0][0])
    for k in range(4):
        assert abs(got[k][0] - ref[k][0]) < 2e-2 * abs(ref[k][0])
