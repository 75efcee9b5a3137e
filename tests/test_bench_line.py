"""bench.py's ONE stdout line stays small enough for the driver to parse (round 5's was 27 KB and was recorded as `parsed: null`).

The canned result (tests/data/bench_result_canned.json) is a full one-GPU result object of round 5 -- eight config lines with per-kernel
counter blocks -- i.e. the input that broke the contract."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _canned():
    with open(os.path.join(ROOT, "tests", "data", "bench_result_canned.json")) as f:
        return json.load(f)


def _strict(text):
    def no_const(name):
        raise ValueError(f"non-strict JSON constant {name}")
    return json.loads(text, parse_constant=no_const)


def test_compact_line_is_small_strict_json_with_the_contract_keys():
    import bench
    res = _canned()
    assert len(json.dumps(res)) > 20000            # the canned input really is the oversized one
    text = bench.compact_line(res, "bench_detail.json")
    assert len(text) < bench.LINE_TARGET_BYTES < bench.LINE_LIMIT_BYTES == 8192
    assert "\n" not in text
    line = _strict(text)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in line, k
    assert line["metric"] == res["metric"] and line["value"] == res["value"] and line["ms_per_step"] == res["ms_per_step"]
    assert set(line["config"]) == {"workload", "sharding"}
    roof = line["roofline"]
    for k in ("kernel", "bound", "achieved", "peak", "unit", "frac", "avg_launch_ms", "algorithmic_bytes_per_launch", "traffic", "step_frac"):
        assert k in roof, k
    assert roof["frac"] == pytest.approx(roof["achieved"] / roof["peak"], rel=1e-3)
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in line["cpu_baseline"], k
    assert "configs" not in line and "kernels_traffic" not in text


def test_compact_line_multi_gpu_shape_and_nan_handling():
    import bench
    res = _canned()
    res.pop("configs"), res.pop("cpu_baseline")
    res["n_gpus"] = 8
    res["comm"] = {"collective": "allreduce/slabs=0", "candidates_ms_per_step": {"allreduce/slabs=0": 1.0, "a2a/slabs=0": None},
                   "collectives_per_step": 4, "bytes_per_collective": 64000000, "compute_alone_ms_per_step": 0.4,
                   "exposed_ms_per_step": float("nan"), "world_size_seen_by_backend": 8, "backend": "nccl", "model": "x" * 5000}
    clean = bench._no_nan(res)
    assert clean["comm"]["exposed_ms_per_step"] is None
    text = bench.compact_line(clean)
    assert len(text) < bench.LINE_TARGET_BYTES
    line = _strict(text)
    assert line["comm"]["world_size_seen_by_backend"] == 8 and "model" not in line["comm"]
    assert "cpu_baseline" not in line


def test_oversized_line_is_refused():
    import bench
    res = _canned()
    res["config"]["workload"] = "w" * 9000
    with pytest.raises(AssertionError):
        bench.compact_line(res)


def test_emit_prints_the_line_last_and_writes_the_detail_file(tmp_path):
    detail = tmp_path / "detail.json"
    code = ("import json, sys; sys.path.insert(0, %r); import bench; "
            "bench.emit(json.load(open(%r)))" % (ROOT, os.path.join(ROOT, "tests", "data", "bench_result_canned.json")))
    env = dict(os.environ, RGCN_BENCH_DETAIL=str(detail))
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    out = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(out) == 1 and len(out[0]) < 8192
    line = _strict(out[0])
    full = _strict(detail.read_text())
    assert full["value"] == line["value"] and len(full["configs"]) == 8
    assert "bench.py detail: " in p.stderr


@pytest.mark.gpu
def test_real_one_gpu_bench_line_parses(tmp_path):
    """the real thing, small sizes: the last stdout line of `python bench.py` is strict JSON < 8 KB with roofline and cpu_baseline"""
    env = dict(os.environ, RGCN_BENCH_DETAIL=str(tmp_path / "d.json"))
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "4", "--warmup", "1", "--no-cpu-s1", "--no-configs",
                        "--sustained-steps", "0"], capture_output=True, text=True, env=env, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    last = [ln for ln in p.stdout.splitlines() if ln.strip()][-1]
    assert len(last) < 8192
    line = _strict(last)
    assert line["n_gpus"] == 1 and line["roofline"]["frac"] > 0 and line["cpu_baseline"]["kind"] == "port"
    assert line["ms_per_step"] > 0 and line["value"] == pytest.approx(10_000_000 / (line["ms_per_step"] * 1e-3), rel=1e-6)
