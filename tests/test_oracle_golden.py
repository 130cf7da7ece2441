"""The Python oracle (oracle/mpc_oracle.py) against vectors produced by the reference
itself (tests/golden/, generator oracle/gen_golden.py).  Tiers P1 and P5 of SURVEY §8c."""
import math

import numpy as np
import pytest

from oracle import mpc_oracle as orc
from tests import util


@pytest.mark.parametrize("n_steps", [3, 8, 32])
def test_g1_objective_and_constraint(n_steps):
    g = util.load("g1_objective.npz")
    k = "n%d_" % n_steps
    params = util.params_from(g["param_keys"], g[k + "params"])
    assert params["control_steps"] == n_steps
    cmap = util.oracle_costmap(g[k + "cells"], g[k + "map_meta"])
    probs = util.problems_from(g[k + "problems"])
    worst = 0.0
    for j in range(len(probs)):
        prob = util.oracle_problem(probs[j], g[k + "footprint"][j])
        f = orc.objective(g[k + "u"][j], prob, params, cmap)
        ref = g[k + "objective"][j]
        worst = max(worst, abs(f - ref) / max(1.0, abs(ref)))
        assert abs(f - ref) <= 1e-12 * max(1.0, abs(ref)), (j, f, ref)
        assert cmap.footprint_cost(prob.footprint) == g[k + "footprint_cost"][j]
        for i in range(n_steps):
            c = orc.f_constraint(g[k + "u"][j], i, params)
            assert abs(c - g[k + "constraint"][j, i]) <= 1e-15
    assert worst <= 1e-12


def test_g1_covers_the_quirks():
    """the fixture must actually exercise the lethal branch, the kink and OOB cells."""
    g = util.load("g1_objective.npz")
    assert (g["n3_objective"] > 300.0).any()          # a 1000*c^2/N lethal term (py:257-258)
    assert (g["n32_footprint_cost"] == 1.0).any()     # w_footprint term active (py:262-263)
    probs = util.problems_from(g["n3_problems"])
    assert (g["n3_u"][0, 0:3] == probs["cur_vel"][0]).all()


def test_g2_yaw():
    g = util.load("g2_yaw.npz")
    for q, rpy in zip(g["q_xyzw"], g["rpy"]):
        assert orc.yaw_from_quaternion(*q) == rpy[2]
    for yaw, q in zip(g["yaws"], g["quat_wxyz"]):
        assert np.allclose(orc.quaternion_from_yaw(yaw), q, rtol=0, atol=1e-16)


@pytest.mark.parametrize("n_steps", [1, 3, 8])
def test_g5_shift(n_steps):
    g = util.load("g5_shift.npz")
    for a, b, r in zip(g["n%d_init" % n_steps], g["n%d_guess" % n_steps], g["n%d_result" % n_steps]):
        assert (orc.initial_guess_update(a.copy(), b.copy(), n_steps) == r).all()


def test_g3_slsqp_restated_objective_reproduces_reference_solves():
    """SciPy SLSQP on the RESTATED objective walks the same iterates as on the
    reference's objective (same SciPy version)."""
    import scipy
    g = util.load("g3_solves.npz")
    if scipy.__version__ not in str(g["versions"]):
        pytest.skip("fixture made with another SciPy")
    params = util.params_from(g["param_keys"], g["params"])
    cmaps = (util.oracle_costmap(np.zeros_like(g["cells"]), g["map_meta"]),
             util.oracle_costmap(g["cells"], g["map_meta"]))
    probs = util.problems_from(g["problems"])
    for j in range(0, 32):
        prob = util.oracle_problem(probs[j])
        r = orc.solve_slsqp(prob, params, cmaps[g["has_map"][j]], np.zeros(9))
        # 1-ulp differences in f are amplified by SciPy's 1.49e-8 forward-difference step,
        # so iterates agree to ~1e-7, not to the last bit.
        assert abs(r.nit - g["nit_loose"][j]) <= 1
        assert np.allclose(r.x, g["x_loose"][j], rtol=0, atol=1e-5)
        assert abs(r.fun - g["f_loose"][j]) <= 1e-8 * max(1.0, abs(r.fun))


@pytest.mark.parametrize("fixture", util.EPISODE_FIXTURES)
def test_g4_wrapper_episodes_with_injected_solver_output(fixture):
    """P5: given the reference's raw solver output, the restated optimizer() wrapper
    reproduces responses and state across the recorded episodes (control_steps 3 and 8)."""
    g = util.load(fixture)
    params = util.params_from(g["param_keys"], g["params"])
    cmap = util.oracle_costmap(g["cells"], g["map_meta"])
    probs = util.problems_from(g["problems"])
    n_ep, n_calls = probs.shape
    for ep in range(n_ep):
        state = orc.ServerState(params["control_steps"])
        for k in range(n_calls):
            prob = util.oracle_problem(probs[ep, k], g["footprint"][ep, k])
            inject = (g["raw_x"][ep, k].copy(), bool(g["success"][ep, k]))
            if k > 0:   # the warm start handed to the solver is the previous call's shift
                assert (state.initial_guess == g["init_guess"][ep, k - 1]).all() or \
                    (g["problems"][ep, k] != g["problems"][ep, k]).any()
            out, _ = orc.optimizer_step(state, prob, params, cmap,
                                        solver=lambda *a, inject=inject: inject)
            assert np.allclose(out, g["out"][ep, k], rtol=0, atol=1e-15), (ep, k)
            assert np.allclose(state.initial_guess, g["init_guess"][ep, k], rtol=0, atol=1e-15)
            assert np.allclose(state.last_control, g["last_control"][ep, k], rtol=0, atol=1e-15)
            assert state.collision == bool(g["collision"][ep, k]), (ep, k)
            assert state.collision_footprint == bool(g["collision_footprint"][ep, k])
            assert abs(state.waiting_time - g["waiting_time"][ep, k]) <= 1e-12


def test_g4_full_episode_with_scipy():
    """Same episodes, but re-solving with SciPy SLSQP on the restated objective.
    Only the first calls are compared: SLSQP at ftol=1e-3 stops on an early iterate
    that is chaotic in 1-ulp perturbations of f (by call 6 the reference path and
    the restated path differ by 1.6e-3 in the command) -- see DESIGN.md "Parity"."""
    import scipy
    g = util.load("g4_episodes.npz")
    if scipy.__version__ not in str(g["versions"]):
        pytest.skip("fixture made with another SciPy")
    params = util.params_from(g["param_keys"], g["params"])
    cmap = util.oracle_costmap(g["cells"], g["map_meta"])
    probs = util.problems_from(g["problems"])
    for ep in (0, 1):
        state = orc.ServerState(3)
        for k in range(5):
            prob = util.oracle_problem(probs[ep, k], g["footprint"][ep, k])
            out, _ = orc.optimizer_step(state, prob, params, cmap)
            assert np.allclose(out, g["out"][ep, k], rtol=0, atol=1e-5), (ep, k)


# ------------------------------------------------------------------ f-4: local_plan (py:271-310)
def _path_from_oracle(x, xy, q, params):
    prob = orc.Problem(xy, q, (0, 0), (0, 0, 0, 1), (0, 0, 0), (0, 0, 0, 1), (0, 0, 0))
    return orc.predicted_path(x, prob, params)


@pytest.mark.parametrize("n_steps", [3, 8, 32])
def test_g7_local_plan(n_steps):
    """`publishLocalPlan` run by the reference on random controls and map->base_link transforms:
    the restated rollout + w-first quaternion reproduce the published Path (positions, and the
    orientation fields exactly as py:301-305 fills them)."""
    g = util.load("g7_local_plan.npz")
    k = "n%d_" % n_steps
    params = util.params_from(g["param_keys"], g[k + "params"])
    for x, xy, q, path in zip(g[k + "x"], g[k + "tf_xy"], g[k + "tf_q"], g[k + "path"]):
        assert (path[0, :2] == xy).all() and (path[0, 2:] == (0.0, 0.0, 0.0, 1.0)).all()   # py:288-291
        got = _path_from_oracle(x, xy, q, params)
        for i, (px, py, yaw) in enumerate(got):
            assert abs(px - path[i + 1, 0]) <= 1e-12 and abs(py - path[i + 1, 1]) <= 1e-12
            w, qx, qy, qz = orc.quaternion_from_yaw(yaw)
            assert np.allclose((qx, qy, qz, w), path[i + 1, 2:], rtol=0, atol=1e-15)


@pytest.mark.parametrize("fixture", util.EPISODE_FIXTURES)
def test_g4_local_plan_of_the_episodes(fixture):
    """the Path the reference published inside optimizer() (py:365 -> 271-310, tf = the request's pose):
    rollout of the UNFILTERED solver output."""
    g = util.load(fixture)
    params = util.params_from(g["param_keys"], g["params"])
    probs = util.problems_from(g["problems"])
    for ep in range(probs.shape[0]):
        for k in range(0, probs.shape[1], 7):
            row = probs[ep, k]
            got = np.array(_path_from_oracle(g["raw_x"][ep, k], row["cur_xy"], row["cur_q"], params))
            ref = g["local_plan"][ep, k]
            assert np.abs(got[:, :2] - ref[1:, :2]).max() <= 1e-12
            assert np.abs(np.sin(0.5 * got[:, 2]) - ref[1:, 4]).max() <= 1e-12   # orientation.z
            assert np.abs(np.cos(0.5 * got[:, 2]) - ref[1:, 5]).max() <= 1e-12   # orientation.w


# ------------------------------------------------------------------ G6: SciPy's FD gradient of the reference objective
@pytest.mark.parametrize("n_steps", [3, 8, 32])
def test_g6_analytic_gradient_of_the_solver_mirror(n_steps):
    """The analytic adjoint gradient (+ control-norm gradient) the build's solver uses == what SLSQP sees
    through approx_derivative on the reference's objective (forward differences, step 1.49e-8):
    agreement to the FD error, ~1e-6 relative to the gradient's scale."""
    import ctypes as C
    from oracle import c_oracle
    from neo_mpc_planner2_amd import abi
    g = util.load("g6_fd_gradient.npz")
    k = "n%d_" % n_steps
    params = util.params_from(g["param_keys"], g[k + "params"])
    probs = util.problems_from(g[k + "problems"])
    u = np.ascontiguousarray(g[k + "u"])
    lib = c_oracle.load()
    ps = abi.params_struct(params)
    zero = (np.zeros((200, 200), np.uint8), 0.05, -5.0, -5.0)
    cells, margs = c_oracle._map_args(zero)
    out = np.zeros_like(u)
    lib.orc_gradient_batch(C.byref(ps), *margs, C.c_void_p(probs.ctypes.data), C.c_void_p(u.ctypes.data),
                           C.c_void_p(out.ctypes.data), C.c_size_t(len(probs)))
    scale = np.abs(g[k + "grad"]).max(axis=1, keepdims=True)
    assert (np.abs(out - g[k + "grad"]) <= 2e-6 * np.maximum(scale, 1e-3)).all(), \
        (np.abs(out - g[k + "grad"]) / np.maximum(scale, 1e-3)).max()


@pytest.mark.parametrize("pset", ["cut", "turn", "readme"])
def test_g8_restated_objective_at_other_parameter_sets(pset):
    """G8: the reference's own objective values at its SLSQP solutions for parameter sets away from the README's
    (the vx/vy box cutting the speed disc; a fast-turning robot, longer horizon, other weights), control_steps 3
    and 8: the restated objective (Python and C) reproduces them."""
    from oracle import c_oracle
    g = util.load("g8_solves_params.npz")
    for n_steps in ((16,) if pset == "readme" else (3, 8)):
        k = "%s_n%d_" % (pset, n_steps)
        params = util.params_from(g["param_keys"], g[k + "params"])
        assert params["control_steps"] == n_steps
        probs = util.problems_from(g[k + "problems"])
        hm = g[k + "has_map"].astype(bool)
        for mask, cells in ((~hm, np.zeros_like(g[k + "cells"])), (hm, g[k + "cells"])):
            cmap_c = (cells,) + tuple(g[k + "map_meta"])
            for tag in ("loose", "tight"):
                f_c = c_oracle.objective_batch(params, cmap_c, probs[mask], g[k + "x_" + tag][mask])
                assert np.allclose(f_c, g[k + "f_" + tag][mask], rtol=1e-12, atol=1e-12)
            cm = util.oracle_costmap(cells, g[k + "map_meta"])
            for j in np.where(mask)[0][:6]:
                f_py = orc.objective(g[k + "x_tight"][j], util.oracle_problem(probs[j]), params, cm)
                assert abs(f_py - g[k + "f_tight"][j]) <= 1e-12 * max(1.0, abs(f_py))
