"""`python bench.py --gpus N` started plainly must itself start N ranks: run it with the gloo
backend switch and the stub workload (a torch-only two-layer net: plumbing, no HIP kernels) and read rank 0's line."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*flags):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "2"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *flags], env=env, capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]                     # ONE JSON line, from rank 0
    return json.loads(lines[0])


def test_gpus_2_starts_two_ranks_by_itself():
    rec = _run("--gpus", "2", "--steps", "3", "--warmup", "1", "--backend", "gloo", "--workload", "stub")
    assert rec["n_gpus"] == 2 and rec["steps"] == 3 and rec["warmup"] == 1
    assert rec["config"]["global_batch"] == 2 and rec["scaling"] == "weak"
    assert rec["value"] > 0 and abs(rec["value"] - 2 * 3 / (rec["ms_per_step"] * 3e-3)) < 1e-6 * rec["value"]


def test_single_rank_plain_start():
    rec = _run("--gpus", "1", "--steps", "2", "--warmup", "0", "--backend", "gloo", "--workload", "stub")
    assert rec["n_gpus"] == 1


def test_rank_count_mismatch_is_refused():
    """Started by a launcher with a world size other than --gpus: refuse instead of reporting a wrong n_gpus."""
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29533")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--workload",
                        "stub"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "ranks" in (r.stderr + r.stdout)
