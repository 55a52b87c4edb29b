"""`python bench.py --gpus N` as a bare command: with WORLD_SIZE unset the script starts N rank processes through
torch.distributed.run before touching the GPU and forwards rank 0's JSON line.  On CPU the ranks run TGR_BENCH_DRYRUN (gloo
rendezvous, barrier + max-reduce protocol, no env); on the GPU box the same launcher runs the real rehearsal (2 ranks sharing cuda:0)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra_env, *args, timeout=600):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(extra_env)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], env=env, capture_output=True, text=True, timeout=timeout)


def test_bare_multi_rank_launch_prints_one_json_line():
    out = _run({"TGR_BENCH_DRYRUN": "1"}, "--gpus", "2", "--steps", "3", "--warmup", "1")
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, out.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["dry_run"] is True and d["value"] is None


def test_a_failing_rank_gives_a_nonzero_exit_code_and_no_line():
    out = _run({"TGR_BENCH_DRYRUN": "0"}, "--gpus", "2", "--steps", "1", "--warmup", "0") if not _has_gpu() else None
    if out is None:
        pytest.skip("needs a box without a HIP device: there the real ranks fail loudly")
    assert out.returncode != 0 and not [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert "HIP device" in out.stderr or "failed" in out.stderr


def _has_gpu():
    import torch
    return torch.cuda.is_available()


@pytest.mark.gpu
def test_bare_two_rank_rehearsal_on_one_gpu():
    """2 ranks sharing cuda:0 over gloo (TGR_BENCH_REHEARSAL): the numbers mean nothing, the path is the 8-GPU one"""
    out = _run({"TGR_BENCH_REHEARSAL": "1"}, "--gpus", "2", "--steps", "20", "--warmup", "5", "--envs-per-gpu", "512", "--no-cpu-baseline")
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["config"]["envs_per_gpu"] == 512 and d["scaling"] == "weak"


@pytest.mark.gpu
def test_the_multi_rank_code_path_runs_over_rccl_on_one_gpu():
    """TGR_BENCH_FORCE_DIST=1: world size 1, but everything the N > 1 path does is done -- a process group over RCCL (backend "nccl"), the
    asynchronous all-gathers of RolloutGatherer every K steps, the barrier and the max-reduce of the elapsed time.  No 8-GPU node is ever
    available to the tests: this is what keeps RCCL and the gather in the driver-run suite (a scaling curve it is not)."""
    out = _run({"TGR_BENCH_FORCE_DIST": "1"}, "--gpus", "1", "--steps", "40", "--warmup", "10", "--envs-per-gpu", "1024", "--no-cpu-baseline", "--no-vecenv")
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["value"] > 0 and d["config"]["gather_every"] is not None
    assert d["config"].get("collective_backend") == "nccl" and d["config"].get("all_gathers_issued", 0) >= 2, d["config"]
