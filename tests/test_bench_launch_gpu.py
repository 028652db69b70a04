"""bench.py's N > 1 path with the real workload: `--gpus 2 --backend gloo` is the rehearsal form (two ranks sharing the
box's one GPU, gloo carrying the gradient buckets): self-launch, PatchParallel with hook-launched exchanges from the
side stream, the timed region's barriers, the per-kernel pass with its collectives, rank 0's line."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_two_rank_rehearsal_on_one_device():
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_PORT="29549")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--workload", "flavr",
                        "--size", "64", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--kernel-steps", "2"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    rec = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert rec["n_gpus"] == 2 and rec["steps"] == 3 and rec["scaling"] == "weak"
    assert rec["config"]["global_batch"] == 2 and "REHEARSAL" in rec["config"]["parallelism"]
    assert rec["value"] > 0 and rec["ms_per_step"] > 0
    assert rec["roofline"]["frac"] > 0                          # the per-kernel pass ran on both ranks (it has collectives)
