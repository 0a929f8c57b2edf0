"""The multi-GPU path of bench.py on hardware, as far as one GPU allows: under the driver's launcher
(`python -m torch.distributed.run --nproc-per-node 1 ...`) the rank initialises RCCL (`nccl`) and both
all-reduces of reduce_totals run on the device; and `--gpus 2` on a box with one GPU fails with a clear message."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALL = ["--steps", "1", "--warmup", "0", "--corpus-jobs", "2048", "--no-c2", "--no-c5", "--no-api", "--no-inflate", "--no-cpu-baseline"]


def _env():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    return env


@pytest.mark.gpu
def test_one_rank_under_the_launcher_runs_the_rccl_collectives():
    import bench
    cmd = bench.launcher_command(1, ["--gpus", "1"] + SMALL, 29400 + os.getpid() % 500)
    p = subprocess.run(cmd, env=dict(_env(), NXZ_BENCH_TRACE_COLLECTIVES="1"), capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line["n_gpus"] == 1 and line["value"] > 0 and line["config"]["parallelism"] == "shard1"
    assert "collectives: backend nccl, all_reduce SUM + MAX on cuda" in p.stderr        # the RCCL path ran, on the device


@pytest.mark.gpu
def test_more_gpus_than_the_box_has():
    import torch
    have = torch.cuda.device_count()
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(have + 1), "--steps", "1"], env=_env(),
                       capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert p.returncode == 2
    assert "%d GPUs requested, %d visible" % (have + 1, have) in p.stderr
