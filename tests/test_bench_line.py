"""The ONE stdout line of bench.py stays small enough for the driver to parse (a 20.8 KB line came back `parsed: null`)."""
import json
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config")


def test_stub_line_is_one_small_json_line(tmp_path):
    detail = tmp_path / "detail.json"
    env = dict(os.environ, PT_BENCH_STUB="1")
    p = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--steps", "2", "--warmup", "1", "--detail-out", str(detail)],
                       capture_output=True, text=True, env=env, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, p.stdout
    assert len(lines[0]) < 8000
    d = json.loads(lines[0])
    for k in CONTRACT:
        assert k in d, k
    assert isinstance(d["config"]["workload"], str) and "model" not in d["config"]
    assert d["steps"] == 2 and d["warmup"] == 1
    full = json.loads(detail.read_text())
    assert full["value"] == d["value"] or abs(full["value"] - d["value"]) < 1e-3 * d["value"]


def test_compact_projection_of_a_full_record_fits():
    """The largest record the bench has produced (round 5: every leg, 20.8 KB) projects to < 6 KB and keeps roofline + cpu_baseline + summary."""
    sys.path.insert(0, REPO)
    import bench
    with open(os.path.join(REPO, "profiles", "r05", "bench_four_stages_final.json")) as f:
        rec = json.load(f)
    line = json.dumps(bench.compact_line(rec), separators=(",", ":"))
    assert len(line) < bench.LINE_LIMIT <= 6000
    d = json.loads(line)
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "launches", "avg_launch_ms"):
        assert k in r, k
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert "frac" in r["det_backbone"] and "frac" in r["det_backbone"]["net_only"]
    assert "conv3x3 implicit GEMM" in r["by_class"]["classes"]
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample", "config0", "parity_sample"):
        assert k in c, k
    assert d["summary"]["pages_per_s_bf16"] == round(rec["value"], 1)
