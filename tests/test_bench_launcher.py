"""bench.py's own rank launcher (no GPU needed for what is checked here): `python bench.py --gpus N` must never exit without
ONE JSON line on stdout — a result, or an "error"."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _clean_env(**extra):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MDH_BENCH_SHARED_GPU")}
    env.update(extra)
    return env


def _json_lines(text):
    return [json.loads(ln) for ln in text.splitlines() if ln.startswith("{")]


def test_more_ranks_than_gpus_is_an_error_line():
    import torch

    n = torch.cuda.device_count() + 2
    run = subprocess.run([sys.executable, "bench.py", "--gpus", str(n), "--steps", "1", "--warmup", "0", "--cells", "8"], cwd=ROOT, env=_clean_env(),
                         capture_output=True, text=True, timeout=300)
    got = _json_lines(run.stdout)
    assert run.returncode != 0 and len(got) == 1
    assert got[0]["n_gpus"] == n and got[0]["value"] is None and "visible GPU" in got[0]["error"]


def test_a_failing_rank_becomes_an_error_line_and_the_others_are_stopped():
    """two ranks are started (shared-GPU switch: no device count check in the launcher); without a GPU each of them fails when it
    selects its device — the launcher must report the first failure with that rank's stderr and exit non-zero"""
    import torch

    if torch.cuda.is_available():
        import pytest

        pytest.skip("the ranks would run; the failing-rank case on a GPU box is in tests/test_gpu_distributed.py")
    run = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--steps", "1", "--warmup", "0", "--cells", "8"], cwd=ROOT,
                         env=_clean_env(MDH_BENCH_SHARED_GPU="1", MDH_BENCH_LAUNCH_TIMEOUT="240"), capture_output=True, text=True, timeout=400)
    got = _json_lines(run.stdout)
    assert run.returncode != 0 and len(got) == 1
    assert got[0]["error"].startswith("rank ") and " of 2 ended with " in got[0]["error"]


def test_world_size_and_gpus_must_agree():
    run = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--steps", "1", "--warmup", "0", "--cells", "8"], cwd=ROOT,
                         env=_clean_env(WORLD_SIZE="3", RANK="0", LOCAL_RANK="0"), capture_output=True, text=True, timeout=300)
    assert run.returncode != 0 and "must agree" in (run.stdout + run.stderr)
