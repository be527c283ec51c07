"""bench.py contract on the GPU box: ONE JSON line, last on stdout, with the fields the driver reads - for the plain
single-GPU step and for the multi-GPU step (RCCL all-gather of the packed per-shard top-k + merge), which
`--force-collective` runs with a single rank so that the N > 1 code is exercised on a 1-GPU machine."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

REQUIRED = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "config", "roofline"}


def _run(*extra):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533")
    out = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--rows", "300000", "--nq", "1024", "--steps", "3", "--warmup", "1", *extra],
                         capture_output=True, text=True, timeout=900, env=env, cwd=str(ROOT))
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    line = json.loads(lines[-1])  # the JSON line is the last line of stdout, whatever the libraries printed before
    assert REQUIRED <= set(line), REQUIRED - set(line)
    assert sum(ln.lstrip().startswith("{") for ln in lines) == 1
    return line


def test_single_gpu_line_with_roofline_and_cpu_baseline():
    line = _run("--cpu-seconds", "2")
    assert line["n_gpus"] == 1 and line["steps"] == 3 and line["value"] > 0 and line["vs_baseline"] is None
    assert set(line["roofline"]) >= {"bound", "achieved", "peak", "unit", "frac", "traffic"}
    assert set(line["cpu_baseline"]) >= {"value", "unit", "cores", "kind", "sample"}
    assert line["verify"]["recall_at_k_vs_torch_fp32"] == 1.0


def test_multi_gpu_step_runs_with_one_rank():
    line = _run("--force-collective", "--no-cpu-baseline")
    assert "all-gather" in line["config"]["parallelism"]
    assert line["verify"]["recall_at_k_vs_torch_fp32"] == 1.0
