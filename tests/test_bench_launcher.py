"""bench.py --gpus N must start its own N ranks (no torchrun) and relay rank 0's JSON line.

CPU-only: DT_BENCH_DRYRUN=1 stops each rank after the rendezvous + rank all-gather (gloo), so this
exercises the launcher, the environment it builds and the relay, not the GPU work.
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, **env):
    e = dict(os.environ, DT_BENCH_DRYRUN="1", **env)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=e, capture_output=True, text=True,
                          timeout=300)


def test_gpus_flag_spawns_that_many_ranks():
    res = _run(["--gpus", "3"])
    assert res.returncode == 0, res.stderr
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 3
    assert [r["rank"] for r in out["ranks"]] == [0, 1, 2]
    assert [r["local_rank"] for r in out["ranks"]] == [0, 1, 2]
    assert len({r["pid"] for r in out["ranks"]}) == 3 and os.getpid() not in {r["pid"] for r in out["ranks"]}


def test_single_rank_needs_no_launcher():
    out = json.loads(_run(["--gpus", "1"]).stdout.strip().splitlines()[-1])
    assert out["n_gpus"] == 1 and len(out["ranks"]) == 1


def test_failing_rank_fails_the_run():
    # without the dry-run switch the ranks refuse to run on a GPU-less host: the parent must report that
    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "DT_BENCH_DRYRUN"):
        e.pop(k, None)
    e["CUDA_VISIBLE_DEVICES"] = e["HIP_VISIBLE_DEVICES"] = ""
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=e, capture_output=True,
                         text=True, timeout=300)
    assert res.returncode != 0
    assert "needs an MI355X" in res.stderr
