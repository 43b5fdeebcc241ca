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


def test_bench_main_multi_rank_prologue_runs_on_gloo():
    """`bench.py --gpus 2` died with UnboundLocalError in main() (bind_rank_to_cores used above its import)
    and --selftest-launch returned before that branch.  This walks the REAL main() with two ranks -- init_process_group,
    bind_rank_to_cores, warm-up, timed loop, pack_matches, all_gather_matches, max-over-ranks, the JSON line -- on gloo with a
    stand-in model (GIM_BENCH_DRY_MODEL=1), the way test.py:188-218 runs its N ranks."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["GIM_BENCH_DRY_MODEL"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1", "--batch", "2"],
                       capture_output=True, text=True, timeout=240, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["n_ranks_seen"] == 2 and len(set(out["rank_devices"])) == 2
    assert out["metric"] == "image-pairs/sec at 640x480" and out["scaling"] == "weak" and out["value"] > 0
    assert out["config"]["pairs_per_step"] == 4 and out["config"]["matches_per_pair"] == 3.5   # (3 + 4) matches per pair over ranks 0, 1
    assert out["steps"] == 4 and abs(out["value"] - 2 * 2 * 4 / (out["ms_per_step"] * 4e-3)) < 0.01 * out["value"]


def test_bench_main_single_rank_dry():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["GIM_BENCH_DRY_MODEL"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--batch", "2"],
                       capture_output=True, text=True, timeout=120, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert out["n_gpus"] == 1 and out["n_ranks_seen"] == 1 and out["config"]["matches_per_pair"] == 3.0
