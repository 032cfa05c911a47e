"""bench.py --gpus N has to be startable exactly like the N = 1 line: without WORLD_SIZE in the environment
it becomes the launcher itself -- one rank per GPU under torch.distributed.run on this node, same command line -- and rank 0
prints the one JSON line.  --dry-launch walks that path without GPUs (gloo), including a real all-reduce over the ranks."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra=None):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout                  # ONE JSON line, from rank 0 only
    return json.loads(lines[0])


def test_gpus_2_without_a_launcher_spawns_its_own_ranks():
    d = _run(["--gpus", "2", "--dry-launch"])
    assert d["dry_launch"] is True and d["n_gpus"] == 2 and d["launcher"] is True
    assert d["allreduce_check"] == d["expected"] == 3.0          # ranks 0 and 1 both took part: 1 + 2


def test_gpus_1_needs_no_launcher():
    d = _run(["--dry-launch"])
    assert d["n_gpus"] == 1 and d["launcher"] is False


def test_launched_by_the_driver_with_torch_distributed_run():
    """The driver's own form: python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N (no second launcher inside)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29563", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-launch"],
                       capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1 and json.loads(lines[0])["n_gpus"] == 2


def test_mismatched_world_size_is_an_error_not_a_silent_single_rank_run():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK")}
    env["WORLD_SIZE"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-launch"], capture_output=True, text=True,
                       timeout=120, env=env, cwd=ROOT)
    assert r.returncode != 0 and "WORLD_SIZE" in (r.stderr + r.stdout)
