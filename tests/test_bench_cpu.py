"""bench.py's own launcher (`python bench.py --gpus N` without torchrun), checked on CPU.

`--launch-check` makes every spawned rank join a gloo process group, all-reduce its rank and exit before any GPU
call, so the N = 2 path - fresh child per rank, RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* exported, rank 0's stdout
is the command's stdout, a failing rank fails the command - is proven to reach `init_process_group` here."""
import json
import os
import subprocess
import sys

from conftest import ROOT


def _bench(*args, env_extra=None):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, str(ROOT / "bench.py"), *args], capture_output=True, text=True, timeout=300, env=env, cwd=str(ROOT))


def test_self_launch_two_ranks_reaches_the_process_group():
    out = _bench("--gpus", "2", "--launch-check")
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.lstrip().startswith("{")]
    assert len(lines) == 1  # only rank 0 reports
    rec = json.loads(lines[0])
    assert rec == {"launch_check": "ok", "world": 2, "rank_sum": 1.0}


def test_external_launcher_world_size_mismatch_is_refused():
    out = _bench("--gpus", "2", "--launch-check", env_extra={"WORLD_SIZE": "3", "RANK": "0"})
    assert out.returncode != 0 and "WORLD_SIZE=3" in (out.stderr + out.stdout)


def test_single_gpu_without_a_gpu_fails_loudly():
    import torch

    if torch.cuda.is_available():
        return
    out = _bench("--steps", "1", "--warmup", "0")
    assert out.returncode != 0 and "needs a GPU" in (out.stderr + out.stdout)


def test_kloop_microbenchmark_cross_compiles(tmp_path):
    """experiments/ubench/kloop.hip (the K loop rebuilt from its parts: profiles/README.md) stays buildable for gfx950."""
    import pathlib
    import shutil

    import pytest

    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not pathlib.Path(hipcc).exists():
        pytest.skip("no hipcc")
    out = subprocess.run([hipcc, "-O1", "-std=c++17", "--offload-arch=gfx950", "-Wno-unused-value", "-c", "-o", str(tmp_path / "kloop.o"),
                          str(ROOT / "experiments" / "ubench" / "kloop.hip")], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]


# ---- round 4: the launcher's failure modes (the driver's 8-GPU run gets one try) ---------------------------------------------


def test_too_few_gpus_is_refused_before_any_rank_starts():
    """`--gpus N` on a node that shows fewer than N GPUs: the parent (which never touches a GPU) refuses with a clear message."""
    out = _bench("--gpus", "4", "--steps", "1", env_extra={"VODHIP_BENCH_FAKE_GPUS": "2"})
    assert out.returncode == 2 and "only 2 GPU(s) visible" in out.stderr
    here = _bench("--gpus", "2", "--steps", "1")  # the real probe: this container has no GPU (on a GPU box the pool has one)
    import torch

    if not torch.cuda.is_available():
        assert here.returncode == 2 and "GPU(s) visible" in here.stderr


def test_a_rank_dying_at_init_fails_the_launch_and_its_stderr_is_reported():
    out = _bench("--gpus", "3", "--launch-check", "--init-timeout", "60", env_extra={"VODHIP_BENCH_TEST_FAULT": "die:1"})
    assert out.returncode == 41
    assert "rank 1 left with exit code 41" in out.stderr and "injected failure before the process-group init" in out.stderr
    assert "---- rank 0" in out.stderr and "---- rank 2" in out.stderr  # every rank's tail is in the one record
    assert not [ln for ln in out.stdout.splitlines() if ln.lstrip().startswith("{")]  # no JSON line from a failed launch


def test_a_rank_that_never_arrives_times_out_instead_of_hanging():
    """Rank 2 hangs before the rendezvous: the others' init watchdog fires after --init-timeout, they leave with code 75, the
    launcher terminates the hung rank and reports - the command ends in seconds, non-zero."""
    import time

    t0 = time.time()
    out = _bench("--gpus", "3", "--launch-check", "--init-timeout", "8", env_extra={"VODHIP_BENCH_TEST_FAULT": "hang:2"})
    assert out.returncode == 75, (out.returncode, out.stderr[-1500:])
    assert "did not complete within 8 s" in out.stderr
    assert time.time() - t0 < 120
