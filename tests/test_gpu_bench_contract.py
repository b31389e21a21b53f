"""bench.py's contract, on the GPU: every workload prints ONE JSON line that carries the fields the driver parses — metric / value / unit /
n_gpus / steps / warmup / ms_per_step / higher_is_better / scaling / vs_baseline / dtype / data / config.workload — plus the `roofline`
object (dominant kernel, algorithmic bytes per launch over its HIP-event duration, against the 8 TB/s peak) and the `cpu_baseline` object
(the oracle timed on the host's cores, on a bounded sample).  Small scale factors: the line's shape is what is tested, not its numbers (at SF1 the
longest kernel of Q3 is a small one)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("workload, metric", [("join", "tpch_q3_hash_join_rows_per_sec"), ("q1", "tpch_q1_rows_per_sec"), ("q3", "tpch_q3_rows_per_sec")])
def test_bench_line_carries_roofline_and_cpu_baseline(workload, metric):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--workload", workload, "--sf", "1", "--cpu-sf", "1", "--steps", "3", "--warmup", "1"]
    p = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=600)
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert p.returncode == 0 and len(lines) == 1, (p.returncode, p.stdout[-2000:], p.stderr[-4000:])
    line = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in line, (k, sorted(line))
    assert line["metric"] == metric and line["n_gpus"] == 1 and line["steps"] == 3 and line["warmup"] == 1 and line["higher_is_better"] is True
    assert line["value"] > 0 and line["ms_per_step"] > 0 and line["vs_baseline"] is None and line["data"] == "synthetic" and "workload" in line["config"]
    r = line["roofline"]
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and r["unit"] == "GB/s" and 0 <= r["frac"] < 1 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3, r
    assert "traffic" in r and r["kernel"]
    c = line["cpu_baseline"]
    assert c["value"] > 0 and c["cores"] >= 1 and c["kind"] in ("port", "reference") and c["sample"] and c["unit"] == line["unit"], c
    if workload in ("q1", "q3"):
        # configs 4 / 5: the CPU leg runs the GPU leg's own tables at its scale factor, and its result is the GPU leg's (the oracle as checker)
        assert c["same_workload_as_gpu_leg"] is True and c["result_equals_gpu_leg"] is True and line["speedup_vs_cpu_port"] > 0, c
        assert line["hbm_frac_whole_step"] == pytest.approx(line["kernel_algorithmic_bytes_per_step"] / (line["ms_per_step"] * 1e-3) / 1e9 / 8000.0, rel=1e-2, abs=2e-4)
        assert line["survey_8d_formula"]["bytes_per_step"] > 0
    if workload == "q3":
        assert c["intermediate_rows_equal_gpu_leg"] is True, c


def test_default_line_also_carries_configs_4_and_5():
    """the default single-GPU line (what the driver runs) carries BASELINE configs 4 and 5 under `also`, each from its own process after the
    headline's timed region: ms_per_step, roofline, whole-step fraction, the CPU leg on the same tables and the oracle's verdict on the result
    (small scale factors here: the shape of the line, not its numbers)"""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--sf", "1", "--cpu-sf", "1", "--steps", "3", "--warmup", "1", "--also", "q1:1,q3:1"]
    p = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=900)
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert p.returncode == 0 and len(lines) == 1, (p.returncode, p.stdout[-2000:], p.stderr[-4000:])
    line = json.loads(lines[0])
    assert line["metric"] == "tpch_q3_hash_join_rows_per_sec" and set(line["also"]) == {"q1", "q3"}
    for w, d in line["also"].items():
        assert "error" not in d, d
        assert d["metric"] == f"tpch_{w}_rows_per_sec" and d["ms_per_step"] > 0 and d["roofline"]["bound"] == "hbm" and 0 < d["hbm_frac_whole_step"] < 1, d
        assert d["cpu_baseline"]["same_workload_as_gpu_leg"] is True and d["cpu_baseline"]["result_equals_gpu_leg"] is True, d["cpu_baseline"]
