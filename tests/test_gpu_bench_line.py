"""bench.py's ONE stdout line as the driver reads it, on the GPU: at most 4096 bytes, strict JSON, the contract's keys,
a live `roofline` and a `cpu_baseline` — VERDICT r4 #1 (round 4's 21.7 KB line was never parsed). A reduced batch and
a short cpu_baseline budget keep this to seconds; the line's size does not depend on either."""
import json
import os
import subprocess
import sys

import pytest

from _common import ROOT

pytestmark = pytest.mark.gpu


def test_the_default_command_prints_one_bounded_strict_line():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1",
                          "--batch", "512", "--cpu-budget", "2"], capture_output=True, text=True, env=env, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, out.stdout[-2000:]
    assert len(lines[0].encode()) <= 4096, len(lines[0])

    def no_constants(c):
        raise AssertionError(f"non-strict JSON constant {c}")
    rec = json.loads(lines[0], parse_constant=no_constants)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in rec, k
    assert rec["n_gpus"] == 1 and rec["steps"] == 3 and rec["warmup"] == 1 and rec["value"] > 0 and rec["dtype"] == "f32"
    assert "workload" in rec["config"] and "model" not in rec["config"]
    r = rec["roofline"]
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and r["peak"] == 157.3 and 0 < r["frac"] < 1
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and r["kernel"].startswith("ins_seg_decode")
    c = rec["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and c["unit"] == "object-crops/s"
    full = json.load(open(os.path.join(ROOT, "gpurun_out", "bench_full.json")))           # the whole table went to the file
    assert full["value"] == rec["value"] and "ins_seg_encode_kernel" in full["kernels"]
