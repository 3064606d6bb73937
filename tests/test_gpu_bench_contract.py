"""bench.py's output contract on a small workload: one JSON line on stdout with the fields the round-end
driver reads, the roofline and CPU-baseline objects, and, for two ranks (a rehearsal on one GPU: the ranks
share the device and exchange through gloo), the auto and the dense leg of the gradient exchange."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALL = ["--gaussians", "200000", "--width", "640", "--height", "368", "--steps", "2", "--warmup", "1", "--no-extras"]


def _run(args):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, cwd=ROOT, env=env, capture_output=True,
                       text=True, timeout=280)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines          # exactly one line on stdout; everything else goes to stderr
    return json.loads(lines[0])


def test_single_gpu_line():
    d = _run(SMALL)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1
    assert d["unit"] == "it/s" and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["vs_baseline"] is None and d["dtype"] == "f32" and "synthetic" in d["data"]
    assert "workload" in d["config"] and "model" not in d["config"]
    assert d["value"] > 0 and abs(d["value"] - 8 * 1e3 / d["ms_per_step"]) < 1e-6 * d["value"]   # 8 views per step
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] in ("hbm", "valu", "mfma")
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and "sample" in c


def test_two_rank_rehearsal_line():
    d = _run(["--gpus", "2", "--no-cpu-baseline"] + SMALL)
    assert d["n_gpus"] == 2 and d["scaling"] == "weak"
    a = d["allreduce"]
    assert a["ranks_seen"] == 2 and a["policy"] == "auto"
    assert a["steps_dense"] + a["steps_sparse"] == 2
    # the leg in which every gradient row travels is there whenever the timed steps did not already send all of them
    if a["steps_sparse"]:
        assert a["dense"] is not None and a["dense"]["steps_dense"] == a["dense"]["steps"] and a["dense"]["it_s"] > 0
    assert abs(d["value"] - 2 * 8 * 1e3 / d["ms_per_step"]) < 1e-6 * d["value"]   # the whole job's views per second
