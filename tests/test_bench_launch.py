"""`python bench.py --gpus N` starts N ranks by itself: the launch / rendezvous / barrier-bracketed timing /
one-JSON-line path rehearsed on CPU (gloo, --dry-run: no prover), and a launcher whose rank count differs from --gpus is refused."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env=None):
    e = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RUN_ID"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, env=e, timeout=600)


def test_gpus_2_spawns_two_ranks_and_prints_one_line():
    r = _run(["--gpus", "2", "--steps", "2", "--dry-run"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.split("\n") if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["ranks"] == [0, 1] and out["steps"] == 2 and out["dry_run"] is True
    # the timed region is the MAX over ranks: rank 1 sleeps twice as long as rank 0
    assert out["ms_per_step"] >= 19.0


def test_one_rank_line_is_unchanged_and_a_mismatched_launcher_is_refused():
    r = _run(["--steps", "3", "--dry-run"])
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([l for l in r.stdout.split("\n") if l.startswith("{")][0])
    assert out["n_gpus"] == 1 and out["ranks"] == [0]
    r = _run(["--gpus", "4", "--dry-run"], env={"WORLD_SIZE": "2", "RANK": "0"})
    assert r.returncode != 0 and "launcher started 2 ranks" in (r.stderr + r.stdout)
