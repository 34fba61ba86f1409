"""bench.py --gpus N started plainly must create its N ranks itself.  Rehearsed here with 2 ranks
on the one GPU of the test box over gloo (RCCL needs one GPU per rank): the JSON line reports the
world that was really initialised, per-rank rates and a checked result gather."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_spawns_its_ranks():
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--frames", "8",
           "--steps", "2", "--warmup", "1", "--no-cpu-baseline"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    try:
        p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=240)
    except (OSError, PermissionError) as e:          # a harness that forbids starting programs from a GPU process
        pytest.skip("could not start bench.py as a child process: %r" % (e,))
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    lines = p.stdout.decode().strip().splitlines()
    assert len(lines) == 1, lines                    # exactly one JSON line on stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2
    d = out["distributed"]
    assert d["world_size_initialised"] == 2 and len(d["per_rank_frames_per_s"]) == 2
    assert d["result_gather"]["frames"] == 16 and d["result_gather"]["counts_and_payload_check"] == "ok"
    assert out["config"]["frames_per_step_whole_job"] == 16
    assert out["value"] > 0 and "invariant" in out


def test_rccl_branch_executes_at_world_size_one():
    """bench.py --force-dist initialises the nccl (= RCCL) process group on the one GPU of the box, so that the
    RCCL initialisation, the timing barrier, the max-over-ranks all-reduce and the device-tensor result gather run
    at least here before a multi-GPU node sees them."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--force-dist", "--frames", "16", "--steps", "2", "--warmup", "1",
           "--no-cpu-baseline", "--no-extras", "--settle-ms", "50"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    try:
        p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=240)
    except (OSError, PermissionError) as e:
        pytest.skip("could not start bench.py as a child process: %r" % (e,))
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    lines = [ln for ln in p.stdout.decode().splitlines() if ln.startswith('{"metric"')]
    assert len(lines) == 1
    out = json.loads(lines[0])
    d = out["distributed"]
    assert d["backend"].startswith("nccl") and d["world_size_initialised"] == 1
    assert d["result_gather"]["frames"] == 16 and d["result_gather"]["counts_and_payload_check"] == "ok"
    assert out["n_gpus"] == 1 and out["value"] > 0
