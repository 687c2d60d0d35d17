"""bench.py prints ONE short JSON line (the driver keeps a bounded tail of stdout: round 5's 25.5 KB line was lost) and writes everything
else to bench_details.json.  CPU: the stub run's line, and compact_line() on a committed full record of a real run."""
import json
import os
import subprocess
import sys

from tests.conftest import ROOT

CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config")


def test_cpu_stub_line_is_short_and_complete(tmp_path):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--cpu-stub", "--pairs", "4"],
                       capture_output=True, text=True, timeout=600, cwd=ROOT, env=dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"))
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1 and r.stdout.rstrip().endswith(lines[0])          # the line is the LAST thing on stdout
    import bench
    assert len(lines[0]) < bench.MAX_LINE_BYTES
    d = json.loads(lines[0])
    assert all(k in d for k in CONTRACT) and isinstance(d["config"]["workload"], str) and "model" not in d["config"]
    assert d["details"] and os.path.exists(os.path.join(ROOT, d["details"]))
    full = json.load(open(os.path.join(ROOT, d["details"])))
    assert full["value"] == d["value"] or abs(full["value"] - d["value"]) <= 1e-5 * abs(full["value"])


def test_compact_line_of_a_full_real_record():
    """the round-5 driver run's full record (25.5 KB, committed as profiles/r05_bench_default_v4.json) through compact_line(): under the limit,
    with the roofline / cpu_baseline objects and one number + one fraction per other configuration"""
    sys.path.insert(0, ROOT)
    import bench
    full = json.load(open(os.path.join(ROOT, "profiles", "r05_bench_default_v4.json")))
    assert len(json.dumps(full)) > 20000
    line = json.dumps(bench.compact_line(full, "bench_details.json"))
    assert len(line) < bench.MAX_LINE_BYTES - 1024               # head-room for a longer workload string / 8 ranks of per-rank numbers
    d = json.loads(line)
    assert all(k in d for k in CONTRACT)
    rf = d["roofline"]
    assert all(k in rf for k in ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "avg_us_per_launch", "work_per_launch", "mfma_busy"))
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-4
    assert all(k in d["cpu_baseline"] for k in ("value", "unit", "cores", "kind", "sample"))
    assert all(k in d["sinkhorn_roofline"] for k in ("frac", "achieved", "traffic"))
    assert d["single_pair"]["ms_per_pair"] > 0
    for name in ("cfg3", "cfg5"):
        assert d["other_configs"][name]["pairs_per_s"] > 0 and 0 < d["other_configs"][name]["frac"] < 1
    assert "per_pair" not in d["ir_fmr_parity"] and "per_pair" not in d["ir_fmr_parity"]["stress_head"]
