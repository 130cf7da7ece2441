"""Random parameter sets, the default search against THE SAME search run to the end (tolerances 1e-10, window rules off, 400
iterations) on all-free maps, where the minimiser is unique up to the control norm's kink: an early stop shows as a distance
between the two first controls.  This is the fuzz that found the one-sided slides and the closing-in rule's false positives
(round 4).  The GPU variant needs no checker at all: K1 against K1."""
import numpy as np
import pytest

from neo_mpc_planner2_amd import synthetic
from tests import util


def random_set(rng):
    n = int(rng.choice([3, 3, 4, 5, 6, 8, 10, 12, 16]))
    vmax = rng.uniform(0.3, 1.2)
    lim = dict(max_vel_x=vmax * rng.uniform(0.5, 1.0), min_vel_x=-vmax * rng.uniform(0.1, 1.0),
               max_vel_y=vmax * rng.uniform(0.2, 1.0), max_vel_trans=vmax, max_vel_theta=rng.uniform(0.4, 1.6))
    lim["min_vel_y"] = -lim["max_vel_y"]
    lim["min_vel_theta"] = -lim["max_vel_theta"] * rng.uniform(0.5, 1.0)
    wt = rng.uniform(0.2, 2.0)
    w = dict(w_trans=wt, w_orient=rng.uniform(0.1, 1.5), w_control=10 ** rng.uniform(-2, -0.2), w_terminal=10 ** rng.uniform(-2, 0),
             w_costmap=wt * rng.uniform(0.01, 0.24), prediction_horizon=rng.uniform(0.4, 1.6),
             opt_tolerance=float(rng.choice([1e-3, 1e-3, 1e-4])))
    return util.orc.make_params(control_steps=n, **{k: float(v) for k, v in lim.items()}, **{k: float(v) for k, v in w.items()})


def run(solve, seed, sets, count):
    rng = np.random.default_rng(seed)
    total = short = 0
    worst = 0.0
    for si in range(sets):
        params = random_set(rng)
        _, cmap, probs, _, _ = synthetic.make_workload("C2", seed=100 + si, batch=count)
        free = (np.zeros_like(cmap[0]),) + tuple(cmap[1:])
        tight = dict(params, window_tolerance=-1.0, step_tolerance=1e-10, cost_tolerance=1e-14, max_iterations=400)
        c1, x1 = solve(params, free, probs)
        c2, x2 = solve(tight, free, probs)
        du = np.abs(x1[:, :3] - x2[:, :3]).max(axis=1)
        assert (c1["status"] == 0).all() and (c1["cost"] <= c2["cost"] + 1e-4).all(), (si, (c1["cost"] - c2["cost"]).max())
        total += count
        short += int((du > 1e-3).sum())
        worst = max(worst, float(du.max()))
    return total, short, worst


def test_default_search_against_run_to_the_end_mirror():
    from oracle import c_oracle

    def solve(params, cmap, pr):
        st, warm = synthetic.make_states(pr, params["control_steps"])
        cm, x, _ = c_oracle.solve_batch(params, cmap, pr, st, warm)
        return cm, x
    total, short, worst = run(solve, seed=1, sets=16, count=128)
    print("run-to-the-end fuzz (mirror): %d solves, %d end more than 1e-3 from the converged first control, worst %.2e" % (total, short, worst))
    assert short <= total // 1000 + 1 and worst <= 5e-3, (short, total, worst)


@pytest.mark.gpu
def test_default_search_against_run_to_the_end():
    from neo_mpc_planner2_amd.solver import BatchSolver

    def solve(params, cmap, pr):
        st, warm = synthetic.make_states(pr, params["control_steps"])
        with BatchSolver(params) as s:
            s.set_costmap(*cmap)
            return s.solve(pr, st, warm)
    total, short, worst = run(solve, seed=1, sets=40, count=256)
    print("run-to-the-end fuzz: %d solves, %d end more than 1e-3 from the converged first control, worst %.2e" % (total, short, worst))
    assert short <= total // 1000 + 1 and worst <= 5e-3, (short, total, worst)
