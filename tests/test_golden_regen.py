"""Keeps the oracle's pin honest: every fixture set is re-derived from the REFERENCE ITSELF (oracle/gen_golden.py imports
/root/reference under the ROS stand-ins) into a scratch directory and must equal tests/golden/ array for array (NaN-aware).
Development container only -- skipped where /root/reference does not exist (the GPU box).  The quick sets (seconds) run in
the default CPU suite; the long ones (cold SLSQP solves at control_steps up to 32, converged episodes: about an hour on 7
cores) with NEO_MPC_REGEN_ALL=1 (`slow`)."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
REFERENCE = "/root/reference/neo_mpc_planner2/mpc_optimization_server.py"

pytestmark = pytest.mark.skipif(not os.path.exists(REFERENCE), reason="the reference is not on this machine")

#: generator name -> the files it writes
QUICK = {"g1": ["g1_objective.npz"], "g2": ["g2_yaw.npz"], "g5": ["g5_shift.npz"], "g6": ["g6_fd_gradient.npz"],
         "g7": ["g7_local_plan.npz"], "g4": ["g4_episodes.npz", "g4_episodes_n8.npz"], "g4b": ["g4_episodes_params.npz"]}
SLOW = {"g3": ["g3_solves.npz"], "g8": ["g8_solves_params.npz"], "g8mid": ["g8_mid.npz"],
        "g9": ["g9_solves_pydefaults.npz", "g9_episodes_pydefaults.npz"], "g3n32": ["g3_solves_n32_zero.npz"],
        "g10": ["g10_heldout.npz"], "g11": ["g11_warm_converged.npz", "g11_warm_converged_n8.npz"],
        "g12": ["g12_after_tuning.npz"], "g13": ["g13_warm_converged_set_a.npz", "g13_warm_converged_set_a_n5.npz"], "g14": ["g14_random_sets.npz"],
        "g15": ["g15_judge_sets.npz"], "g16": ["g16_judge_sets_r5.npz"],
        "g17": ["g17_warm_costmap_sets.npz"], "g18": ["g18_held_out_sets.npz"]}


def _regenerate_and_compare(name, files, tmp_path):
    env = dict(os.environ, NEO_MPC_GOLDEN_OUT=str(tmp_path))
    env.pop("NEO_FUZZ_CACHE", None)   # (from the reference, not from answers a fuzz run left behind)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "oracle", "gen_golden.py"), name], env=env,
                         capture_output=True, text=True)
    assert out.returncode == 0, out.stderr[-2000:]
    for f in files:
        new, old = np.load(os.path.join(str(tmp_path), f)), np.load(os.path.join(GOLDEN, f))
        assert sorted(new.files) == sorted(old.files), f
        for k in old.files:
            if k == "versions":
                assert str(new[k]) == str(old[k]), "fixture %s was made with %s, this container has %s" % (f, old[k], new[k])
            elif old[k].dtype.kind in "fc":
                assert new[k].shape == old[k].shape and np.array_equal(new[k], old[k], equal_nan=True), (f, k)
            else:
                assert np.array_equal(new[k], old[k]), (f, k)


def test_every_fixture_file_has_a_generator():
    listed = {f for files in list(QUICK.values()) + list(SLOW.values()) for f in files}
    assert listed == {f for f in os.listdir(GOLDEN) if f.endswith(".npz")}


@pytest.mark.parametrize("name", sorted(QUICK))
def test_quick_sets_regenerate_bit_for_bit(name, tmp_path):
    _regenerate_and_compare(name, QUICK[name], tmp_path)


@pytest.mark.slow
@pytest.mark.skipif(os.environ.get("NEO_MPC_REGEN_ALL") != "1", reason="about an hour of SLSQP: NEO_MPC_REGEN_ALL=1")
@pytest.mark.parametrize("name", sorted(SLOW))
def test_slow_sets_regenerate_bit_for_bit(name, tmp_path):
    _regenerate_and_compare(name, SLOW[name], tmp_path)
