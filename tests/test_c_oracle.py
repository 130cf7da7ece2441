"""The plain-C oracle (oracle/mpc_oracle.c) against the reference's golden vectors and
against the Python restatement; and the CPU mirror of the build's solver against SciPy
SLSQP solves of the reference (tiers P2/P3 of SURVEY §8c)."""
import numpy as np
import pytest

from neo_mpc_planner2_amd import abi, synthetic
from oracle import c_oracle
from oracle import mpc_oracle as orc
from tests import util


@pytest.mark.parametrize("n_steps", [3, 8, 32])
def test_g1_objective(n_steps):
    g = util.load("g1_objective.npz")
    k = "n%d_" % n_steps
    params = util.params_from(g["param_keys"], g[k + "params"])
    cmap = (g[k + "cells"],) + tuple(g[k + "map_meta"])
    probs = util.problems_from(g[k + "problems"]).copy()
    probs["footprint_cost"] = g[k + "footprint_cost"]
    f = c_oracle.objective_batch(params, cmap, probs, g[k + "u"])
    ref = g[k + "objective"]
    assert np.all(np.abs(f - ref) <= 1e-12 * np.maximum(1.0, np.abs(ref)))
    # the rasterised footprint cost itself
    fp = g[k + "footprint"]
    has = ~np.isnan(fp).any(axis=(1, 2))
    fc = c_oracle.footprint_cost_batch(cmap, fp[has])
    assert (fc == g[k + "footprint_cost"][has]).all()


def test_g2_yaw():
    g = util.load("g2_yaw.npz")
    for q, rpy in zip(g["q_xyzw"], g["rpy"]):
        assert c_oracle.yaw(*q) == rpy[2]


@pytest.mark.parametrize("fixture", util.EPISODE_FIXTURES)
def test_g4_wrapper_episodes_injected(fixture):
    """P5 for the C wrapper: responses and state over the recorded episodes, reference x.x injected."""
    g = util.load(fixture)
    params = util.params_from(g["param_keys"], g["params"])
    cmap = (g["cells"],) + tuple(g["map_meta"])
    probs = util.problems_from(g["problems"])
    n_ep, n_calls = probs.shape
    states, warm = abi.new_states(n_ep, params["control_steps"])
    for k in range(n_calls):
        fp = g["footprint"][:, k]
        rows = probs[:, k].copy()
        has = ~np.isnan(fp).any(axis=(1, 2))
        rows["footprint_cost"] = 0.0
        if has.any():
            rows["footprint_cost"][has] = c_oracle.footprint_cost_batch(cmap, fp[has])
        cmds, _, _ = c_oracle.postprocess_batch(params, cmap, rows, states, warm, g["raw_x"][:, k],
                                               g["success"][:, k].astype(np.int32))
        assert np.allclose(cmds["vel"], g["out"][:, k], rtol=0, atol=1e-15), k
        assert np.allclose(warm, g["init_guess"][:, k], rtol=0, atol=1e-15), k
        assert np.allclose(states["last_control"], g["last_control"][:, k], rtol=0, atol=1e-15)
        assert (states["collision"] == g["collision"][:, k]).all(), k
        assert (states["collision_footprint"] == g["collision_footprint"][:, k]).all(), k
        assert np.allclose(states["waiting_time"], g["waiting_time"][:, k], rtol=0, atol=1e-12)
        assert ((cmds["flags"] & abi.FLAG_RESET) != 0).tolist() == \
            [k == 0 or (k == 25 and ep % 2 == 1) for ep in range(n_ep)]
    assert warm.shape[1] == 3 * params["control_steps"]


def _g3(n_steps=3):
    g = util.load("g3_solves.npz")
    k = "" if n_steps == 3 else "n%d_" % n_steps
    params = util.params_from(g["param_keys"], g[k + "params"])
    assert params["control_steps"] == n_steps
    probs = util.problems_from(g[k + "problems"])
    hm = g[k + "has_map"].astype(bool)
    return {name[len(k):]: g[name] for name in g.files if name.startswith(k) or not k}, params, probs, hm


# (control_steps, method): 0 auto (dense Newton at 3, Riccati sweep otherwise), 1 L-BFGS, 2 dense Newton, 3 Riccati
CASES = [(3, 0), (3, 1), (3, 3), (8, 0), (8, 1), (8, 2), (32, 0), (32, 1)]


def _cold_solve(params, cmap, probs):
    st, warm = synthetic.make_states(probs, params["control_steps"])
    cmds, x, _ = c_oracle.solve_batch(params, cmap, probs, st, warm)
    return cmds, x


@pytest.mark.parametrize("n_steps,method", CASES)
def test_p2_solver_mirror_matches_tight_slsqp_where_unique(n_steps, method):
    """P2: zero costmap (unique minimiser): first control within 1e-3 of SciPy SLSQP at
    ftol=1e-12 run on the REFERENCE's objective; objective not worse."""
    g, params, probs, hm = _g3(n_steps)
    params["method"] = method
    if method == 1 and n_steps > 8:
        params["max_iterations"] = 600
    zero = (np.zeros_like(g["cells"]),) + tuple(g["map_meta"])
    cmds, x = _cold_solve(params, zero, probs[~hm])
    ok = g["status_tight"][~hm] == 0       # SLSQP itself arrived (it runs into maxiter 500 at control_steps 32)
    du0 = np.abs(x[:, :3] - g["x_tight"][~hm][:, :3]).max(axis=1)
    assert du0[ok].max() <= 1e-3, du0[ok].max()
    assert du0[ok].max() <= 2e-4            # what the algorithm actually achieves
    assert (cmds["cost"] <= g["f_tight"][~hm] + 1e-8).all()   # never above SLSQP's value, converged or not
    assert (cmds["status"] == 0).all()


@pytest.mark.parametrize("n_steps,method", CASES)
def test_p3_solver_mirror_not_worse_than_reference_tolerance(n_steps, method):
    """P3: all cases incl. costmaps: f(build) <= f(SciPy @ ftol=1e-3) + 1e-3, feasible."""
    g, params, probs, hm = _g3(n_steps)
    params["method"] = method
    if method == 1 and n_steps > 8:
        params["max_iterations"] = 600
    for mask, cells in ((~hm, np.zeros_like(g["cells"])), (hm, g["cells"])):
        cmap = (cells,) + tuple(g["map_meta"])
        cmds, x = _cold_solve(params, cmap, probs[mask])
        assert (cmds["cost"] <= g["f_loose"][mask] + 1e-3).all()
        # cost reported == the reference objective at the returned point
        pr = probs[mask].copy()
        f = c_oracle.objective_batch(params, cmap, pr, x)
        assert np.allclose(f, cmds["cost"], rtol=1e-12, atol=1e-12)
        assert (np.abs(x.reshape(len(x), -1, 3)[:, :, 0]) <= params["max_vel_x"] + 1e-12).all()
        assert (np.abs(x.reshape(len(x), -1, 3)[:, :, 2]) <= params["max_vel_theta"] + 1e-12).all()
        speed = np.hypot(x.reshape(len(x), -1, 3)[:, :, 0], x.reshape(len(x), -1, 3)[:, :, 1])
        assert (speed <= params["max_vel_trans"] + 1e-9).all()


def test_riccati_direction_equals_the_dense_newton_direction():
    """The stage-wise (Riccati) sweep solves the same projected Newton system as the dense elimination:
    next to a minimiser (positive definite, no pivot replaced) both directions agree to the
    finite-difference error of the dense Hessian."""
    import ctypes as C
    lib = c_oracle.load()
    rng = np.random.default_rng(0)
    for n in (3, 8):
        params = orc.make_params(control_steps=n)
        ps = abi.params_struct(params)
        cmap = (np.zeros((200, 200), np.uint8), 0.05, -5.0, -5.0)
        cells, margs = c_oracle._map_args(cmap)
        probs = synthetic.make_problems(40, 200, seed=5)
        st, warm = synthetic.make_states(probs, n)
        _, xs, _ = c_oracle.solve_batch(params, cmap, probs, st, warm)
        for j in range(40):
            u = xs[j] + rng.normal(0, 3e-3, size=3 * n)
            b = u.reshape(n, 3)
            b[:, :2] *= np.minimum(1, 0.7 / np.hypot(b[:, 0], b[:, 1]))[:, None]
            b[:, 2] = np.clip(b[:, 2], -0.7, 0.7)
            dd, ds, dw = np.zeros(3 * n), np.zeros(3 * n), np.zeros(3 * n)
            lib.orc_debug_newton_directions(C.byref(ps), *margs, C.c_void_p(probs[j:j + 1].ctypes.data),
                                            C.c_void_p(u.ctypes.data), C.c_void_p(dd.ctypes.data),
                                            C.c_void_p(ds.ctypes.data), C.c_void_p(dw.ctypes.data))
            assert np.abs(dd - ds).max() <= 1e-5 * max(1e-9, np.abs(dd).max()), (n, j)
            # the device's formulation (displacement coordinates, stage solved in the coordinates of its face)
            assert np.abs(dw - ds).max() <= 1e-9 * max(1e-9, np.abs(ds).max()), (n, j)


def test_projection_box_cuts_disc():
    """box ∩ disc projection when the box cuts the disc (SURVEY §7 hard part 3): the solver
    output must satisfy both and not be worse than SciPy tight."""
    params = orc.make_params(max_vel_trans=0.7, max_vel_x=0.4, min_vel_x=-0.2, max_vel_y=0.65,
                             min_vel_y=-0.65)
    probs = synthetic.make_problems(24, 200, seed=91)
    zero = (np.zeros((200, 200), np.uint8), 0.05, -5.0, -5.0)
    cmds, x = _cold_solve(params, zero, probs)
    xs = x.reshape(len(x), -1, 3)
    assert (xs[:, :, 0] <= 0.4 + 1e-12).all() and (xs[:, :, 0] >= -0.2 - 1e-12).all()
    assert (np.hypot(xs[:, :, 0], xs[:, :, 1]) <= 0.7 + 1e-9).all()
    cm = orc.Costmap(*zero)
    for j in range(8):
        r = orc.solve_slsqp(util.oracle_problem(probs[j]), params, cm, np.zeros(9), ftol=1e-12, maxiter=500)
        assert cmds["cost"][j] <= r.fun + 1e-8
        assert np.abs(x[j, :3] - r.x[:3]).max() <= 1e-3


def test_window_tolerance_only_shortens_creeping_searches():
    """The windowed stop (three iterations gaining < 3e-3 * opt_tolerance together, Newton only by
    default) never lengthens a search, keeps the objective within 1e-4 of the run without it and
    leaves the zero-costmap (unique minimiser) solutions untouched."""
    cmap = synthetic.make_costmap(500, seed=41)
    probs = synthetic.make_problems(1024, 500, seed=42)
    res = {}
    for wt in (-1.0, 0.0):
        res[wt] = _cold_solve(orc.make_params(window_tolerance=wt), cmap, probs)[0]
    off, on = res[-1.0], res[0.0]
    assert (on["iterations"] <= off["iterations"]).all()
    assert on["iterations"].max() < off["iterations"].max()
    assert (on["cost"] <= off["cost"] + 1e-4).all()
    zero = (np.zeros_like(cmap[0]),) + tuple(cmap[1:])
    z_off = _cold_solve(orc.make_params(window_tolerance=-1.0), zero, probs[:256])[1]
    z_on = _cold_solve(orc.make_params(window_tolerance=0.0), zero, probs[:256])[1]
    assert np.abs(z_on[:, :3] - z_off[:, :3]).max() <= 2e-4    # the command (first control)
    assert np.abs(z_on - z_off).max() <= 1e-3
    # L-BFGS (method = 1): off by default
    p12 = orc.make_params(control_steps=12, method=1)
    a = _cold_solve(p12, cmap, probs[:128])[0]
    b = _cold_solve(orc.make_params(control_steps=12, method=1, window_tolerance=-1.0), cmap, probs[:128])[0]
    assert (a["iterations"] == b["iterations"]).all()


def _free_space_workload(n_steps, count, seed):
    """Zero costmap (every rollout in free space): unique minimisers, the trial step is eligible everywhere."""
    cmap = synthetic.make_costmap(300, seed=seed)
    zero = (np.zeros_like(cmap[0]),) + tuple(cmap[1:])
    probs = synthetic.make_problems(count, 300, seed=seed + 1)
    return util.orc.make_params(control_steps=n_steps), zero, probs


def test_damping_beyond_8_control_steps_cuts_the_long_tail():
    """Levenberg-Marquardt damping of the Riccati direction (orc_pg_solve): long horizons need about nine
    iterations instead of a dozen with a tail of dozens; the first control still sits where a solve run to
    the end puts it (the flat-problem drift check of tools/parity_report.py, on the mirror)."""
    for n_steps, mean_bar in ((32, 10.5), (64, 12.5)):
        params, zero, probs = _free_space_workload(n_steps, 256, seed=5)
        cmds, x = _cold_solve(params, zero, probs)
        assert (cmds["status"] == 0).all()
        assert cmds["iterations"].mean() <= mean_bar, cmds["iterations"].mean()
        assert np.percentile(cmds["iterations"], 99) <= 25
        tight = dict(params, window_tolerance=-1.0, step_tolerance=1e-9, cost_tolerance=1e-12, max_iterations=300)
        cmds_t, x_t = _cold_solve(tight, zero, probs)
        assert np.abs(x[:, :3] - x_t[:, :3]).max() <= 4e-4
        assert (cmds["cost"] - cmds_t["cost"]).max() <= 1e-5


def test_trial_step_changes_the_path_not_the_answer():
    """In free space the full Newton step is tried before the 64-candidate search: with the trial switched off
    the same problems end at the same minimisers (1e-3 on every control of the first block, objective 1e-5)."""
    lib = c_oracle.load()
    for n_steps in (8, 32):
        params, zero, probs = _free_space_workload(n_steps, 256, seed=9)
        cmds, x = _cold_solve(params, zero, probs)
        lib.orc_set_trial(0)
        try:
            cmds_s, x_s = _cold_solve(params, zero, probs)
        finally:
            lib.orc_set_trial(1)
        assert np.abs(x[:, :3] - x_s[:, :3]).max() <= 1e-3
        assert np.abs(cmds["cost"] - cmds_s["cost"]).max() <= 1e-5
        assert (cmds["status"] == 0).all() and (cmds_s["status"] == 0).all()


def test_trial_step_needs_free_space():
    """With a costmap term under the rollout the search always runs: the trial changes nothing there.  A map of
    raw cost 1 everywhere (constant term, same minimisers as the zero map) makes every rollout ineligible."""
    lib = c_oracle.load()
    params, zero, probs = _free_space_workload(8, 64, seed=13)
    ones = (np.ones_like(zero[0]),) + tuple(zero[1:])
    cmds, x = _cold_solve(params, ones, probs)
    lib.orc_set_trial(0)
    try:
        cmds_s, x_s = _cold_solve(params, ones, probs)
    finally:
        lib.orc_set_trial(1)
    assert np.array_equal(x, x_s) and np.array_equal(cmds["iterations"], cmds_s["iterations"])


def _g8(pset, n_steps):
    g = util.load("g8_solves_params.npz")
    k = "%s_n%d_" % (pset, n_steps)
    params = util.params_from(g["param_keys"], g[k + "params"])
    assert params["control_steps"] == n_steps
    return {name[len(k):]: g[name] for name in g.files if name.startswith(k)}, params, util.problems_from(g[k + "problems"]), \
        g[k + "has_map"].astype(bool)


@pytest.mark.parametrize("pset,n_steps,method", [("cut", 3, 0), ("cut", 8, 0), ("cut", 8, 2), ("cut", 3, 1),
                                                 ("turn", 3, 0), ("turn", 8, 0), ("turn", 8, 2), ("turn", 3, 1), ("turn", 3, 2),
                                                 ("readme", 16, 0), ("readme", 16, 1)])
def test_p2_p3_solver_mirror_at_other_parameter_sets(pset, n_steps, method):
    """G8: P2 / P3 against the reference's SLSQP solves for parameter sets that take the general code paths (box
    cutting the disc, v_cur outside the feasible set, heading beyond pi/4 within the horizon)."""
    g, params, probs, hm = _g8(pset, n_steps)
    params["method"] = method
    if method == 1 and n_steps > 8:
        params["max_iterations"] = 600
    for mask, cells in ((~hm, np.zeros_like(g["cells"])), (hm, g["cells"])):
        cmap = (cells,) + tuple(g["map_meta"])
        cmds, x = _cold_solve(params, cmap, probs[mask])
        worse = cmds["cost"] - g["f_loose"][mask]
        # (the direction with the wall model: no outliers; AUTO picks it at control_steps 3 too when w_costmap > w_trans / 4)
        riccati = method == 3 or (method == 0 and (n_steps != 3 or params["w_costmap"] > 0.25 * params["w_trans"]))
        if cells.any() and pset == "turn" and not riccati:
            # w_costmap = 0.3 (six times the README's) and lethal cells next to the path: the dense-Newton and
            # L-BFGS directions have no wall model (oracle: orc_wall_model; device: costmap.h) -- FORCED onto this
            # weight (AUTO never sends them there; G8 "mid" pins its threshold) a search blocked by a lethal cell
            # creeps up to it and ends there, in 2 of 24 cases at control_steps 3 (3e-3 and 9e-3 above SLSQP's value,
            # 4e-2 with L-BFGS) and 1 of 12 at 8 (0.68 above); 11 end more than 1e-3 BELOW it.  Hard bounds on the
            # known outliers; the stage-wise direction passes the plain bar.  DESIGN.md section 1.
            assert (worse > 1e-3).sum() <= (2 if n_steps == 3 else 1), np.sort(worse)[-3:]
            assert worse.max() <= (1.0 if n_steps == 8 else 5e-2 if method == 1 else 1.5e-2), worse.max()
        else:
            assert (worse <= 1e-3).all(), worse.max()                                                          # P3
        assert (cmds["status"] == 0).all()
        if not cells.any():
            ok = g["status_tight"][mask] == 0
            du0 = np.abs(x[:, :3] - g["x_tight"][mask][:, :3]).max(axis=1)
            assert du0[ok].max() <= 1e-3, du0[ok].max()                                                        # P2
            assert (cmds["cost"] <= g["f_tight"][mask] + 1e-6).all()


# ------------------------------------------------------------------ round 3: the remaining pins of the solver
def _command(params, cmap, probs, x):
    """the velocity command optimizer() returns for raw solver output x (py:365-395: low-pass + clamp), cold state"""
    st, warm = synthetic.make_states(probs, params["control_steps"])
    return c_oracle.postprocess_batch(params, cmap, probs, st, warm, x.copy())[0]["vel"]


@pytest.mark.parametrize("fixture,prefix", util.G9_GROUPS)
@pytest.mark.parametrize("method", [0, 2, 1])
def test_g9_node_defaults_p2_p3_and_the_literal_command_gate(fixture, prefix, method):
    """G9: the parameter values the node itself declares (py:49-75: opt_tolerance 1e-5, every weight 0.5, w_footprint
    2000, limits 0.5, horizon 0.5).  P2 / P3 as for G3 -- P3 against SLSQP as shipped at THIS tolerance (ftol 1e-5) --
    and the north star's sentence taken literally, on the COMMAND (after low-pass and clamp), zero map:
      (L1) within 1e-3 of the SciPy path run to convergence (ftol 1e-12), every case;
      (L2) never further from the SciPy path as shipped (ftol 1e-5) than that path is from its own converged answer,
           + 1e-4 (SLSQP at 1e-5 still stops up to 0.1 short in u0: the objective is that flat);
      (L3) within 1e-3 of the path as shipped wherever the path as shipped is itself converged to 1e-4."""
    g, params, probs, hm = util.solve_group(fixture, prefix)
    assert params["opt_tolerance"] == 1e-5 and params["w_costmap"] == 0.5 and params["w_footprint"] == 2000
    params["method"] = method
    for mask, cells in ((~hm, np.zeros_like(g["cells"])), (hm, g["cells"])):
        cmap = (cells,) + tuple(g["map_meta"])
        cmds, x = _cold_solve(params, cmap, probs[mask])
        assert (cmds["status"] == 0).all()
        worse = cmds["cost"] - g["f_loose"][mask]
        if cells.any() and method != 0:
            # the directions without hop candidates (dense Newton, L-BFGS; AUTO takes the stage-wise one at this costmap
            # weight): two cases sit a millimetre from a cheaper cell SLSQP's line search happened to land in
            assert (worse <= 1e-3).sum() >= len(worse) - 2 and worse.max() <= 4e-3, np.sort(worse)[-3:]
        else:
            assert (worse <= 1e-3).all(), worse.max()                                                         # P3
        if not cells.any():
            ok = g["status_tight"][mask] == 0
            du0 = np.abs(x[:, :3] - g["x_tight"][mask][:, :3]).max(axis=1)
            assert du0[ok].max() <= 1e-3 and du0[ok].max() <= 2e-4, du0[ok].max()                            # P2
            assert (cmds["cost"] <= g["f_tight"][mask] + 1e-5).all()
            v_build = _command(params, cmap, probs[mask], x)
            v_loose = _command(params, cmap, probs[mask], g["x_loose"][mask])
            v_tight = _command(params, cmap, probs[mask], g["x_tight"][mask])
            d_bt = np.abs(v_build - v_tight).max(axis=1)
            d_bl = np.abs(v_build - v_loose).max(axis=1)
            d_lt = np.abs(v_loose - v_tight).max(axis=1)
            assert d_bt[ok].max() <= 1e-3 and d_bt[ok].max() <= 1e-4, d_bt[ok].max()                          # L1
            assert (d_bl[ok] <= d_lt[ok] + 1e-4).all(), (d_bl - d_lt)[ok].max()                               # L2
            settled = ok & (d_lt <= 1e-4)
            assert settled.sum() >= len(ok) // 2 and d_bl[settled].max() <= 1e-3, d_bl[settled].max()        # L3


@pytest.mark.parametrize("fixture,prefix", util.G8_MID_GROUPS)
@pytest.mark.parametrize("method", [0, 2, 3])
def test_g8_mid_costmap_weights_across_the_auto_threshold(fixture, prefix, method):
    """G8 "mid": the README's parameters with w_costmap / w_trans = 0.10 ... 0.30, every case on the costmap: P3 for
    the dense-Newton direction (AUTO below 1/4) and the stage-wise one (AUTO above) on every case -- the threshold of
    neo_mpc_capi.cpp derive() keeps neither away from problems it cannot do."""
    g, params, probs, hm = util.solve_group(fixture, prefix)
    assert hm.all() and abs(params["w_costmap"] / params["w_trans"] - int(prefix[1:3]) / 100.0) < 1e-12
    params["method"] = method
    cmap = (g["cells"],) + tuple(g["map_meta"])
    cmds, x = _cold_solve(params, cmap, probs)
    assert (cmds["status"] == 0).all()
    assert (cmds["cost"] <= g["f_loose"] + 1e-3).all(), (cmds["cost"] - g["f_loose"]).max()
    assert np.allclose(c_oracle.objective_batch(params, cmap, probs, g["x_tight"]), g["f_tight"], rtol=1e-12, atol=1e-12)


@pytest.mark.parametrize("method", [0, 1])
def test_g3_n32_unique_minimisers(method):
    """G3 at control_steps 32 on the all-free map: 64 problems, SLSQP's maxiter raised until ftol 1e-12 reports
    status 0 (57 of them): P2 for BASELINE config 5 rests on these."""
    g, params, probs, _ = util.solve_group("g3_solves_n32_zero.npz", "")
    assert params["control_steps"] == 32
    params["method"] = method
    if method == 1:
        params["max_iterations"] = 600
    zero = (np.zeros((200, 200), np.uint8),) + tuple(g["map_meta"])
    cmds, x = _cold_solve(params, zero, probs)
    ok = g["status_tight"] == 0
    assert ok.sum() >= 40
    du0 = np.abs(x[:, :3] - g["x_tight"][:, :3]).max(axis=1)
    assert du0[ok].max() <= 1e-3 and du0[ok].max() <= (2e-4 if method == 0 else 5e-4), du0[ok].max()
    assert (cmds["cost"] <= g["f_tight"] + 1e-6).all() and (cmds["cost"] <= g["f_loose"] + 1e-3).all()
    assert (cmds["status"] == 0).all()
    f_at = c_oracle.objective_batch(params, zero, probs, g["x_tight"])
    assert np.allclose(f_at, g["f_tight"], rtol=1e-12, atol=1e-12)


def test_hop_candidates_never_hurt_and_find_the_cheaper_cell():
    """The hop lanes of the stage-wise direction (orc_hops / costmap.h): switched off -- and the cell scan behind the search
    (orc_cell_scan / cell_scan.h, round 5) with them --, the same problems end at the same or a higher objective, and the
    two G9 cases that sit a millimetre from a cheaper cell end 3.5e-3 and 1.7e-3 above the reference's SLSQP value.  The
    cell scan alone (hop lanes off) rescues both as well: its candidates include the hop's."""
    lib = c_oracle.load()
    g, params, probs, hm = util.solve_group("g9_solves_pydefaults.npz", "n3_")
    cmap = (g["cells"],) + tuple(g["map_meta"])
    on = _cold_solve(params, cmap, probs[hm])[0]
    lib.orc_set_hops(0)
    try:
        scan_only = _cold_solve(params, cmap, probs[hm])[0]
        lib.orc_set_scan(0)
        off = _cold_solve(params, cmap, probs[hm])[0]
    finally:
        lib.orc_set_hops(1)
        lib.orc_set_scan(1)
    assert ((off["cost"] - g["f_loose"][hm]) > 1e-3).sum() == 2 and ((on["cost"] - g["f_loose"][hm]) <= 1e-3).all()
    assert ((scan_only["cost"] - g["f_loose"][hm]) <= 1e-3).all()
    assert (on["cost"] <= off["cost"] + 1e-9).mean() >= 0.95


def test_blocked_run_rule_only_shortens_creeping_searches():
    """The blocked-run stop rule of the dense Newton direction (three iterations in a row not won by a decent Newton
    step, gaining < 0.1 x opt_tolerance together -- 0.03 x in free space): cold solves end at the same objective (1e-4)
    in no more iterations, zero-map first controls stay where a solve run to the end puts them, and the warm ticks of a
    closed control loop (warm start = the previous solution shifted by one control step, py:397-400) lose the creeping
    searches that set a launch's duration."""
    lib = c_oracle.load()
    cmap = synthetic.make_costmap(500, seed=0)
    probs = synthetic.make_problems(2048, 500, seed=1000)
    params = orc.make_params(method=2)     # (the dense direction for every instance: AUTO hands walls to the stage-wise one)
    zero = (np.zeros_like(cmap[0]),) + tuple(cmap[1:])
    res = {}
    for on in (0, 1):
        lib.orc_set_blocked_rule(on)
        try:
            res[on] = (_cold_solve(params, cmap, probs)[0], util.closed_loop_on_the_mirror(params, cmap, probs, 24),
                       _cold_solve(params, zero, probs[:1024])[1])
        finally:
            lib.orc_set_blocked_rule(1)
    (c0, l0, z0), (c1, l1, z1) = res[0], res[1]
    assert (c1["iterations"] <= c0["iterations"]).all() and (c1["cost"] <= c0["cost"] + 1e-4).all()
    max0 = np.array([t["iterations"].max() for t in l0[5:]]), np.array([t["iterations"].max() for t in l1[5:]])
    max0, max1 = max0
    assert np.median(max1) <= 15 and max1.max() <= 20 and np.median(max0) >= 18, (max0, max1)
    assert all((t["status"] == 0).all() for t in l1)
    tight = dict(params, window_tolerance=-1.0, step_tolerance=1e-9, cost_tolerance=1e-12, max_iterations=400)
    zt = _cold_solve(tight, zero, probs[:1024])[1]
    assert np.abs(z1[:, :3] - zt[:, :3]).max() <= 1e-3 and np.abs(z1[:, :3] - z0[:, :3]).max() <= 1e-4


# ------------------------------------------------------------------ round 4: held-out sets and the warm gate
@pytest.mark.parametrize("name,n_steps", util.G10_GROUPS)
def test_g10_held_out_parameter_sets(name, n_steps):
    """G10: three parameter sets that were never looked at while thresholds were tuned (two of them the round-3 judge's),
    control_steps 3 / 5 / 8 / 12, 300 x 300 maps of other seeds: P3 on every case, P2 <= 3e-4 on the all-free-map cases."""
    m = util.check_held_out_group(_cold_solve, name, n_steps)
    print("G10 %s N=%d (mirror): P2 %.2e, P3 margin free %.2e map %.2e, iterations %.1f / %.1f"
          % (name, n_steps, m["p2"], m["p3_free"], m["p3_map"], m["it_free"], m["it_map"]))


@pytest.mark.parametrize("fixture", util.G11_FIXTURES)
def test_g11_warm_commands_against_the_converged_reference(fixture):
    """G11: the deployed (warm-started) mode.  The build's command at the README tolerance within 1e-3 of the reference's
    CONVERGED command on >= 99.9 % of the ticks (round 3: 97.8 %)."""
    def solve(params, cmap, rows, st, wm):
        cm, x, _ = c_oracle.solve_batch(params, cmap, rows, st, wm)
        return cm, x

    def post(params, cmap, rows, st, wm, x, success):
        c_oracle.postprocess_batch(params, cmap, rows, st, wm, x, success)
    dv, du, its, settled = util.warm_gate(solve, post, fixture)
    print("G11 %s (mirror): %d ticks, |command diff| p99 %.2e max %.2e, above 1e-3: %d; |u0 diff| above 1e-3: %d; iterations %.2f"
          % (fixture, dv.size, np.percentile(dv, 99), dv.max(), (dv > 1e-3).sum(), (du > 1e-3).sum(), its.mean()))
    assert (dv <= 1e-3).mean() >= 0.999, ((dv > 1e-3).sum(), dv.size, dv.max())
    assert dv.max() <= 3e-3


@pytest.mark.parametrize("name,n_steps", util.G12_GROUPS)
def test_g12_parameter_sets_drawn_after_the_tuning_stopped(name, n_steps):
    """G12: G10's protocol at three more parameter sets and control_steps 3 / 4 / 6 / 10, generated AFTER the last change
    of round 4 to the search or to a threshold (nothing was adjusted on it): P3 on every case, P2 <= 3e-4 where the
    fixture flags the minimiser unique."""
    m = util.check_held_out_group(_cold_solve, name, n_steps, fixture="g12_after_tuning.npz")
    print("G12 %s N=%d (mirror): P2 %.2e over %d unique cases (%d not unique: %.2e), P3 margin free %.2e map %.2e, iterations %.1f / %.1f"
          % (name, n_steps, m["p2"], m["p2_cases"], m["not_unique"], m["p2_not_unique"], m["p3_free"], m["p3_map"], m["it_free"], m["it_map"]))


@pytest.mark.parametrize("fixture", util.G13_FIXTURES)
def test_g13_warm_gate_at_another_parameter_set(fixture):
    """G13: G11's protocol at G10's set "a" (heavy control weight, box cutting the disc), control_steps 3 and 5, generated
    after the tuning stopped."""
    def solve(params, cmap, rows, st, wm):
        cm, x, _ = c_oracle.solve_batch(params, cmap, rows, st, wm)
        return cm, x

    def post(params, cmap, rows, st, wm, x, success):
        c_oracle.postprocess_batch(params, cmap, rows, st, wm, x, success)
    dv, du, its, settled = util.warm_gate(solve, post, fixture)
    above, above_settled, unsettled = util.assert_warm_gate(dv, settled, fixture)
    print("G13 %s (mirror): %d ticks, |command diff| p99 %.2e max %.2e, above 1e-3: %d (%d on settled ticks; the reference's own "
          "answers disagree on %d ticks); iterations %.2f"
          % (fixture, dv.size, np.percentile(dv, 99), dv.max(), above, above_settled, unsettled, its.mean()))


@pytest.mark.parametrize("fixture", sorted(util.RANDOM_SETS))
def test_random_parameter_sets_have_no_misses_mirror(fixture):
    """G14 (the fuzz's first 48 draws) and G15 (the 64 seeds the round-4 judge drew, 9000-9063): exact gates."""
    m = util.random_sets_miss_rates(_cold_solve, fixture)
    print(fixture, "(mirror):", m)
    util.assert_random_sets(m, fixture)


#: the costmap cases of G16 on which round 5's AUTO (the dense direction for every instance at control_steps 3) ended more than
#: 1e-3 above the reference's SLSQP: (seed, case) -- dense searches hemmed in by lethal cells
G16_DENSE_MISSES = {(30027, 17), (61020, 19), (61027, 3), (62024, 7), (62024, 15)}


def test_g16_has_teeth_the_dense_direction_alone_misses_the_wall_cases():
    """G16 with method = NEWTON (the dense direction for every instance: round 5's AUTO at control_steps 3): exactly the five
    known objective misses, all at control_steps 3, all with a lethal cell in the instance's reach tile -- and AUTO (direction
    by neighbourhood) solves every one of them through the stage-wise direction."""
    def dense(params, cmap, pr):
        return _cold_solve(dict(params, method=2) if params["control_steps"] == 3 and params["w_costmap"] <= 0.25 * params["w_trans"] else params, cmap, pr)
    m = util.random_sets_miss_rates(dense, "g16_judge_sets_r5.npz")
    got = {(seed, case) for kind, seed, case, _ in m["misses"] if kind == "P3"}
    assert got == G16_DENSE_MISSES and m["p3_miss_free"] == 0 and m["p2_miss"] == 0, m["misses"]
    g = util.load("g16_judge_sets_r5.npz")
    for seed, case in sorted(G16_DENSE_MISSES):
        grp = {k[len("s%d_" % seed):]: g[k] for k in g.files if k.startswith("s%d_" % seed)}
        params = util.params_from(g["param_keys"], grp["params"])
        assert params["control_steps"] == 3
        cmap = (grp["cells"],) + tuple(grp["map_meta"])
        pr = util.problems_from(grp["problems"])[case:case + 1]
        assert c_oracle.route_batch(params, cmap, pr)[0] == 1
        cm, _ = _cold_solve(params, cmap, pr)
        assert cm["cost"][0] <= grp["f_loose"][case] + 1e-3 and (cm["flags"][0] & abi.FLAG_WALL_IN_REACH)


def test_g16_stagewise_direction_pinned_at_control_steps_3():
    """The round-5 review ran its 105 sets with method = RICCATI forced: no objective miss, and ONE first-control miss on an
    all-free map (seed 62022 / case 0, 1.3e-3): a block in the corner between the speed disc and the vy bound, left a
    rounding error inside the bound by the projection, was not seen as ON the bound, slid along the disc into it, got pinned
    by the corner re-pin and never tried the slide along the bound (bounds are active within NEO_RULE_CORNER_ROOM since).
    The control_steps-3 sets of G16, stage-wise direction for every instance: exact gates."""
    m = util.random_sets_miss_rates(_cold_solve, "g16_judge_sets_r5.npz", only_steps=3, over=dict(method=3))
    print("G16, control_steps 3, method RICCATI (mirror):", {k: v for k, v in m.items() if k != "misses"})
    assert m["cases_free"] >= 300 and m["p3_miss_free"] == 0 and m["p3_miss_map"] == 0 and m["p2_miss"] == 0, m["misses"]
    assert m["p2_worst"] <= 2e-4


def test_direction_by_neighbourhood_on_the_benchmark_workload():
    """AUTO at control_steps 3 (solver_rules.h neo_rules_routes_by_neighbourhood): an instance with no lethal cell in its
    reach tile gets the dense direction's answer bit for bit (method = NEWTON on the same instance), one with a wall in reach
    the stage-wise direction's -- NEO_MPC_FLAG_WALL_IN_REACH says which; 12 % of the config-2 instances.  Nothing is routed
    at other control_steps, at a heavy costmap weight, or with a pinned method."""
    cfg, cmap, probs, st, warm = synthetic.make_workload("C2", seed=0, batch=2048)
    params = orc.make_params()
    routed = c_oracle.route_batch(params, cmap, probs).astype(bool)
    assert 0.08 <= routed.mean() <= 0.16
    auto, xa = _cold_solve(params, cmap, probs)
    dense, xd = _cold_solve(dict(params, method=2), cmap, probs)
    assert ((auto["flags"] & abi.FLAG_WALL_IN_REACH) != 0).tolist() == routed.tolist()
    assert (xa[~routed] == xd[~routed]).all() and (auto["cost"][~routed] == dense["cost"][~routed]).all()
    assert (xa[routed] != xd[routed]).any(axis=1).mean() >= 0.9          # (another search: another path to the answer)
    # ... and where both directions end in the same basin they end at the same objective
    d = auto["cost"][routed] - dense["cost"][routed]
    assert np.median(np.abs(d)) <= 1e-4 and (d <= 1e-3).mean() >= 0.97, (np.median(np.abs(d)), (d <= 1e-3).mean())
    for other in (dict(control_steps=4), dict(w_costmap=0.3), dict(method=3)):
        p2 = orc.make_params(**other)
        pr = probs[:256]
        a2, x2 = _cold_solve(p2, cmap, pr)
        lib = c_oracle.load()
        lib.orc_set_route(0)
        try:
            b2, y2 = _cold_solve(p2, cmap, pr)
        finally:
            lib.orc_set_route(1)
        assert (x2 == y2).all(), other


def test_warm_drift_gate_mirror():
    """12 warm ticks of 4096 robots, then the README-tolerance answer against the same search run to the end: no command
    beyond 1e-3 (util.assert_warm_drift; the GPU test of the same name runs K1)."""
    def solve(params, cmap, p, st, warm):
        cm, x, _ = c_oracle.solve_batch(params, cmap, p, st, warm)
        return cm, x
    du, dv, df, it1, it2 = util.warm_drift(solve)
    print("warm drift (mirror): |u0 diff| max %.2e above 1e-3: %d; |command diff| max %.2e above 1e-3: %d; f diff max %.2e; iterations %.2f vs %.2f"
          % (du.max(), (du > 1e-3).sum(), dv.max(), (dv > 1e-3).sum(), df.max(), it1.mean(), it2.mean()))
    util.assert_warm_drift(du, dv, df)


def test_one_sided_slides_and_the_closing_in_rule_mirror():
    print("stop-rule regressions (mirror): (|du0| of the instance, max over 256, iterations)", util.check_stop_rule_regressions(_cold_solve))


@pytest.mark.parametrize("fixture", util.EPISODE_FIXTURES)
def test_p3_on_the_reference_warm_starts_mirror(fixture):
    """P3w on the CPU mirror (the GPU test of the same name runs K1): every call of the recorded episodes solved from the
    REFERENCE's own state on the real costmap -- f(build) <= f(reference's raw x.x) + 1e-3."""
    g = util.load(fixture)
    params = util.params_from(g["param_keys"], g["params"])
    n = params["control_steps"]
    cmap = (g["cells"],) + tuple(g["map_meta"])
    probs = util.problems_from(g["problems"])
    n_ep, n_calls = probs.shape
    states, warm = abi.new_states(n_ep, n)
    worse = []
    for k in range(n_calls):
        fp = g["footprint"][:, k]
        rows = probs[:, k].copy()
        has = ~np.isnan(fp).any(axis=(1, 2))
        rows["footprint_cost"] = 0.0
        if has.any():
            rows["footprint_cost"][has] = c_oracle.footprint_cost_batch(cmap, fp[has])
        cm, x, _ = c_oracle.solve_batch(params, cmap, rows, states.copy(), warm.copy())
        worse.append(cm["cost"] - c_oracle.objective_batch(params, cmap, rows, g["raw_x"][:, k]))
        assert (cm["status"] == 0).all()
        c_oracle.postprocess_batch(params, cmap, rows, states, warm, g["raw_x"][:, k], g["success"][:, k])
    worse = np.array(worse)
    assert worse.max() <= 1e-3, worse.max()


def test_g17_p3_on_warm_starts_at_random_parameter_sets_mirror():
    """G17 (round 6): P3w away from the four recorded episode files -- 16 random parameter sets x 4 episodes x 30 calls of the
    reference as shipped on the costmap, every call solved from the reference's own state: f(build) <= f(reference's raw
    x.x) + 1e-3 on EVERY call (the GPU test of the same name runs K1)."""
    def solve(params, cmap, rows, st, wm):
        cm, x, _ = c_oracle.solve_batch(params, cmap, rows, st, wm)
        return cm, x

    def post(params, cmap, rows, st, wm, x, ok):
        c_oracle.postprocess_batch(params, cmap, rows, st, wm, x, ok)
    rows = util.p3w_random_sets(solve, post)
    for r in rows:
        print("G17 seed %d control_steps %2d: max f - f_ref %.2e, %d of %d calls above 1e-3, iterations %.2f" % r)
    assert len(rows) == 16 and sum(r[3] for r in rows) == 0 and sum(r[4] for r in rows) == 16 * 4 * 30, rows
