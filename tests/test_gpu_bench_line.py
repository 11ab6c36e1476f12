"""bench.py as the driver starts it: `python bench.py --gpus N --steps K --warmup W` prints exactly
one JSON line on stdout.  N = 1 runs in the process itself; N = 2 must start its own two ranks
(children created before any GPU call) -- rehearsed here with both ranks on the one GPU of the test
box and the staged (gloo) gradient exchange, the same Trainer code path as RCCL apart from the
transport."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, timeout=280):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True,
                       timeout=timeout, env=env, cwd=ROOT)
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    return r, lines


SCHEMA = {"metric": str, "value": float, "unit": str, "n_gpus": int, "steps": int, "warmup": int, "ms_per_step": float,
          "higher_is_better": bool, "scaling": str, "dtype": str, "data": str, "config": dict, "roofline": dict,
          "timing": dict, "devices": list}


def _check_line(out, n, steps, warmup):
    for k, t in SCHEMA.items():
        assert k in out and isinstance(out[k], t), (k, out.get(k))
    assert out["n_gpus"] == n and out["steps"] == steps and out["warmup"] == warmup
    assert out["vs_baseline"] is None and out["scaling"] == "weak" and out["dtype"] == "bf16"
    assert "workload" in out["config"] and out["config"]["global_batch"] == n * out["config"]["batch_per_gpu"]
    rf = out["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "mfma_busy_frac", "eff_clock_ghz"):
        assert k in rf
    assert rf["bound"] == "mfma" and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
    tm = out["timing"]
    assert tm["blocks"] == 3 and len(tm["ms_per_step_host_clock"]) == 3 and len(tm["ms_per_step_hip_events"]) == 3
    assert tm["ms_per_step_min"] <= out["ms_per_step"] <= tm["ms_per_step_max"]
    assert len(out["devices"]) == n and not out["nan_or_inf"]
    per_step = out["config"]["global_batch"] / (out["ms_per_step"] * 1e-3)
    assert abs(per_step - out["value"]) <= 2e-3 * out["value"]


@pytest.mark.gpu
def test_single_gpu_line_schema():
    r, lines = _run(["--gpus", "1", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"])
    assert r.returncode == 0, r.stderr[-2000:]
    assert len(lines) == 1, lines
    out = json.loads(lines[0])
    _check_line(out, 1, 2, 1)
    assert "with_loader" in out and "comm_exposed_ms" not in out


@pytest.mark.gpu
def test_two_ranks_are_started_by_bench_itself():
    r, lines = _run(["--gpus", "2", "--shared-gpu", "--steps", "3", "--warmup", "1", "--no-cpu-baseline"])
    assert r.returncode == 0, r.stderr[-3000:]
    assert len(lines) == 1, lines
    out = json.loads(lines[0])
    _check_line(out, 2, 3, 1)
    assert out["shared_gpu_rehearsal"] is True and out["rccl_ranks"] == 0
    assert out["comm_transport"].startswith("gloo") and out["comm_exposed_ms"] > 0.0


def test_launcher_failure_is_an_exit_code_not_a_line():
    """No GPUs (or fewer than asked for): non-zero exit code, nothing on stdout."""
    import torch
    if torch.cuda.device_count() >= 64:
        pytest.skip("needs a host with fewer than 64 GPUs")
    r, lines = _run(["--gpus", "64", "--steps", "1", "--warmup", "0"], timeout=120)
    assert r.returncode != 0 and lines == []
