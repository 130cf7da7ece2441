"""world_size > 1 on the hardware there is: `bench.py --gpus 2` with both ranks on device 0
(NEO_MPC_BENCH_SHARE_DEVICE=1; RCCL refuses two ranks on one device, so the collective is gloo, host-staged) -- the spawned
ranks, per-rank seeds, the all-gather on device tensors, the MAX reduce over ranks, `rccl.per_rank` and the JSON line all
run with world_size = 2 before the first real multi-GPU run does."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _bench(*args):
    env = dict(os.environ, NEO_MPC_BENCH_SHARE_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--no-cpu-baseline", "--no-pcie",
                          "--no-others"] + list(args), env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    return json.loads(out.stdout.strip().splitlines()[-1])


def test_two_ranks_on_one_device_gather_every_ranks_commands():
    d = _bench("--steps", "6", "--warmup", "2")
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["steps"] == 6
    r = d["rccl"]
    assert r["world_size"] == 2 and r["shared_device"] and len(r["per_rank"]) == 2
    assert [p["rank"] for p in r["per_rank"]] == [0, 1]
    # gathered == the concatenation of the two ranks' OWN commands (different seeds: different slices)
    assert r["gather_check"]["ok"] and r["gather_check"]["per_rank"] == [True, True] and r["gather_check"]["distinct_slices"]
    assert r["gather_bytes_per_rank"] == 4096 * 24
    # whole-job value = the instances of both ranks over the MAX of the ranks' times
    slowest = max(p["ms_per_step"] for p in r["per_rank"])
    assert abs(d["ms_per_step"] - slowest) <= 1e-6 * slowest
    assert abs(d["value"] - 2 * 4096 / (1e-3 * d["ms_per_step"])) <= 1e-6 * d["value"]
    assert d["solver"]["converged_frac"] == 1.0 and "valu_issue" in d and d["roofline"]["hbm"]["bound"] == "hbm"


def test_c4_shards_split_as_design_section_6_says():
    """--workload C4 --batch 8192: every rank solves ITS 8192 instances of the C2 problem (block partition, no data-path
    collective), the all-gather moves 8192 x 24 bytes per rank, the job is 16 384 instances per step."""
    d = _bench("--workload", "C4", "--batch", "8192", "--steps", "4", "--warmup", "1")
    assert d["n_gpus"] == 2 and "batch 8192 instances/GPU" in d["config"]["workload"] and "control_steps=3" in d["config"]["workload"]
    assert "sharded x2" in d["config"]["parallelism"]
    r = d["rccl"]
    assert r["gather_bytes_per_rank"] == 8192 * 24 and r["gather_check"]["ok"]
    assert abs(d["value"] - 2 * 8192 / (1e-3 * d["ms_per_step"])) <= 1e-6 * d["value"]
    assert all(abs(p["value"] - 8192 / (1e-3 * p["ms_per_step"])) <= 1e-6 * p["value"] for p in r["per_rank"])
