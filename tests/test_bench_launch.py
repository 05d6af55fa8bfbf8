"""`python bench.py --gpus N` must launch its own ranks: the spawn path on CPU ranks.

`--dry-run` runs the launcher, the gloo rendezvous on 127.0.0.1, the barrier, the MAX / SUM reductions and the
all_gather of the N > 1 path without solving anything (no GPU here; nothing stands in for the solve and the line says
`"dry_run": true, "value": null`). What a multi-GPU run adds to the single-GPU one is exactly this plumbing."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(extra_env=None, gpus=2):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env.update(extra_env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(gpus), "--dry-run", "--batch", "64",
                           "--steps", "2", "--warmup", "1"], env=env, capture_output=True, text=True, timeout=300)


def test_plain_invocation_spawns_the_ranks_and_prints_one_line():
    r = run()
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout            # exactly ONE line on stdout: rank 0's
    line = json.loads(lines[0])
    assert line["dry_run"] is True and line["value"] is None
    assert line["n_gpus"] == 2 and line["scaling"] == "weak"
    assert line["gather"]["ranks_seen"] == 2 and line["gather"]["distinct_rank_slices"] == 2
    assert line["gather"]["first_scene_id_per_rank"] == [0.0, 64.0]      # weak shards: disjoint scene ids
    assert line["summary"]["scenes"] == 128 and line["summary"]["max_rank"] == 1.0   # SUM and MAX over ranks
    assert line["elapsed_max_s"] >= 0.02                                  # the slower rank's time, not rank 0's


def test_a_failing_rank_fails_the_command():
    r = run({"SMPC_BENCH_DRY_FAIL_RANK": "1"})
    assert r.returncode != 0
    assert r.stdout.strip() == ""               # no line for a run that did not complete on every rank


def test_single_rank_dry_run_needs_no_rendezvous():
    r = run(gpus=1)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip())
    assert line["n_gpus"] == 1 and line["gather"]["ranks_seen"] == 1
