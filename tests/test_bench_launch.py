"""bench.py's launch contract on CPU: `python bench.py --gpus N` without a launcher starts N ranks itself (a child
`torch.distributed.run`, before any GPU call) and reports n_gpus = N; a launcher whose WORLD_SIZE disagrees with --gpus is
refused.  MDT_BENCH_DRYRUN=1 replaces the GPU by CPU tensors over gloo with the compute injected (bench.py DryRunEngine):
what is under test is the launch, the sharding, the gather and the line -- the N > 1 path of SURVEY.md 8e that no
GPU test can reach on a one-GPU box."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALL = ["--steps", "2", "--warmup", "1", "--draws", "40", "--chains", "2", "--params", "6"]


def _run(args, env_extra, timeout=300):
    env = dict(os.environ, MDT_BENCH_DRYRUN="1", **env_extra)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        if k not in env_extra:
            env.pop(k, None)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True,
                          timeout=timeout, cwd=ROOT)


def _line(stdout):
    lines = [ln for ln in stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, stdout  # rank 0 prints ONE JSON line
    return json.loads(lines[0])


@pytest.mark.timeout(600)
def test_gpus_2_without_a_launcher_runs_two_ranks():
    r = _run(["--gpus", "2"] + SMALL, {})
    assert r.returncode == 0, r.stderr[-2000:]
    res = _line(r.stdout)
    assert res["n_gpus"] == 2 and res["dry_run"] is True
    assert res["steps"] == 2 and res["warmup"] == 1 and res["scaling"] == "weak"
    assert res["config"]["sharding"] == "params axis x2" and res["config"]["params_per_gpu"] == 6
    # whole-job aggregate: both ranks' parameters over the slowest rank's time
    assert res["value"] == pytest.approx(2 * 6 * 2 / (res["ms_per_step"] * 1e-3 * 2), rel=1e-9)
    assert "roofline" not in res  # a dry run measures nothing


def test_gpus_1_dry_run_line():
    r = _run(["--gpus", "1"] + SMALL, {})
    assert r.returncode == 0, r.stderr[-2000:]
    res = _line(r.stdout)
    assert res["n_gpus"] == 1 and res["dry_run"] is True and res["config"]["sharding"] == "params axis x1"


def test_world_size_that_disagrees_with_gpus_is_refused():
    r = _run(["--gpus", "2"] + SMALL, {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "refusing" in r.stderr
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


def test_no_gpu_and_no_dry_run_fails_loudly():
    env = {k: v for k, v in os.environ.items() if k not in ("MDT_BENCH_DRYRUN", "WORLD_SIZE", "RANK", "LOCAL_RANK")}
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible: the product path would run")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + SMALL, env=env, capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert r.returncode != 0 and "needs a GPU" in r.stderr
