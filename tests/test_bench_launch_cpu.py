"""`python bench.py --gpus N` must start N ranks by itself (the flag was parsed and ignored).  CPU check of
that launcher path: the real re-exec under torch.distributed.run on 127.0.0.1, with gloo ranks walking the bench's
barrier / timed loop / all-gather(v) / max-over-ranks / one-JSON-line protocol on a stand-in step."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_gpus2_spawns_two_ranks():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--selftest-launch", "--steps", "3"],
                       capture_output=True, text=True, timeout=240, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout  # rank 0 prints ONE line
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["ranks"] == [0, 1] and out["local_ranks"] == [0, 1] and out["distinct_processes"]
    assert out["matches"] == out["expected_matches"] == 3 * (3 + 4)


def test_world_size_mismatch_is_refused():
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--selftest-launch"],
                       capture_output=True, text=True, timeout=120, env=env, cwd=ROOT)
    assert r.returncode != 0 and "WORLD_SIZE=2" in (r.stderr + r.stdout)
