"""GPU parity tests: the HIP path through the C-ABI (libneo_mpc.so) against
  * the reference's golden vectors (tests/golden/),
  * the CPU oracle (oracle/mpc_oracle.c, oracle/mpc_oracle.py) on the same seeded inputs.
Tolerances: 1e-12 relative on objective values (f64 on both sides, FMA contraction only);
1e-3 on velocity commands vs SciPy (north star: the repo's own opt_tolerance); the
GPU-vs-CPU-mirror comparisons of the same algorithm are much tighter (stated per test).
"""
import numpy as np
import pytest

from neo_mpc_planner2_amd import abi, synthetic
from tests import util

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def solver_mod():
    from neo_mpc_planner2_amd import solver
    return solver


def _solver(solver_mod, params, cmap):
    s = solver_mod.BatchSolver(params)
    s.set_costmap(*cmap)
    return s


# ------------------------------------------------------------------ P1: objective
@pytest.mark.parametrize("n_steps", [3, 8, 32])
def test_objective_kernel_matches_reference_golden(solver_mod, n_steps):
    g = util.load("g1_objective.npz")
    k = "n%d_" % n_steps
    params = util.params_from(g["param_keys"], g[k + "params"])
    cmap = (g[k + "cells"],) + tuple(g[k + "map_meta"])
    probs = util.problems_from(g[k + "problems"]).copy()
    probs["footprint_cost"] = g[k + "footprint_cost"]
    with _solver(solver_mod, params, cmap) as s:
        f = s.objective(probs, g[k + "u"])
    ref = g[k + "objective"]
    rel = np.abs(f - ref) / np.maximum(1.0, np.abs(ref))
    assert rel.max() <= 1e-12, rel.max()


# ------------------------------------------------------------------ P5: wrapper episodes
@pytest.mark.parametrize("fixture", util.EPISODE_FIXTURES)
def test_postprocess_kernel_reproduces_reference_episodes(solver_mod, fixture):
    from oracle import c_oracle
    g = util.load(fixture)
    params = util.params_from(g["param_keys"], g["params"])
    cmap = (g["cells"],) + tuple(g["map_meta"])
    probs = util.problems_from(g["problems"])
    n_ep, n_calls = probs.shape
    states, warm = abi.new_states(n_ep, params["control_steps"])
    with _solver(solver_mod, params, cmap) as s:
        for k in range(n_calls):
            fp = g["footprint"][:, k]
            rows = probs[:, k].copy()
            has = ~np.isnan(fp).any(axis=(1, 2))
            rows["footprint_cost"] = 0.0
            if has.any():   # episodes with a footprint: give the polygon's cost (oracle raster)
                rows["footprint_cost"][has] = c_oracle.footprint_cost_batch(cmap, fp[has])
            cmds = s.postprocess(rows, states, warm, g["raw_x"][:, k], g["success"][:, k])
            assert np.allclose(cmds["vel"], g["out"][:, k], rtol=0, atol=1e-14), k
            assert np.allclose(warm, g["init_guess"][:, k], rtol=0, atol=1e-14), k
            assert np.allclose(states["last_control"], g["last_control"][:, k], rtol=0, atol=1e-14)
            assert (states["collision"] == g["collision"][:, k]).all(), k
            assert (states["collision_footprint"] == g["collision_footprint"][:, k]).all(), k
            assert np.allclose(states["waiting_time"], g["waiting_time"][:, k], rtol=0, atol=1e-12)


def test_footprint_raster_on_device_matches_oracle(solver_mod):
    """polygons handed to the kernel (lanes rasterise edges) == the oracle's Bresenham."""
    from oracle import c_oracle
    footprint_world = synthetic.footprint_world
    cmap = synthetic.make_costmap(200, seed=5)
    probs = synthetic.make_problems(512, 200, seed=6)
    fps = np.array([footprint_world(r) for r in probs])
    want = c_oracle.footprint_cost_batch(cmap, fps)
    params = util.orc.make_params(w_footprint=2000)
    st, warm = synthetic.make_states(probs, 3)
    with _solver(solver_mod, params, cmap) as s:
        cmds, x = s.solve(probs, st, warm, footprints=fps)
    assert (st["collision_footprint"] == (want == 1.0)).all()
    assert (want == 1.0).any() and (want != 1.0).any()
    assert (cmds["vel"][want == 1.0] == 0.0).all()


# ------------------------------------------------------------------ solver vs CPU mirror
@pytest.mark.parametrize("n_steps,count,map_size,method", [(3, 1024, 500, 0), (3, 1024, 500, 1), (8, 256, 200, 0),
                                                           (32, 64, 200, 0), (8, 256, 200, 1),
                                                           # run-time-sized dense Newton kernel (control_steps <= 8)
                                                           (8, 512, 300, 2), (5, 512, 300, 2), (1, 256, 300, 2),
                                                           # Riccati sweep (auto picks it for control_steps != 3)
                                                           (3, 512, 300, 3), (12, 256, 300, 0), (32, 64, 200, 1),
                                                           (64, 32, 200, 0),
                                                           # damped Riccati direction, trial step (beyond 8 steps)
                                                           (16, 256, 300, 0), (32, 512, 500, 0)])
def test_solver_kernel_matches_cpu_mirror(solver_mod, n_steps, count, map_size, method):
    """Same algorithm, same inputs, f64 on both sides: GPU vs oracle/mpc_oracle.c.
    Differences come only from sincos/atan2 implementations, FMA contraction and the
    summation order of the wave reductions."""
    from oracle import c_oracle
    # L-BFGS at control_steps=32 needs more than SLSQP's 100 iterations (96 variables, memory 4)
    # method 0 = auto: dense projected Newton at control_steps 3, Riccati sweep otherwise; 1 = L-BFGS;
    # 2 = dense Newton (control_steps <= 8); 3 = Riccati sweep
    params = util.orc.make_params(control_steps=n_steps, max_iterations=100 if (n_steps < 32 or method != 1) else 600,
                                  method=method)
    cmap = synthetic.make_costmap(map_size, seed=11)
    probs = synthetic.make_problems(count, map_size, seed=12 + n_steps)
    st_g, warm_g = synthetic.make_states(probs, n_steps)
    st_c, warm_c = synthetic.make_states(probs, n_steps)
    with _solver(solver_mod, params, cmap) as s:
        cg, xg = s.solve(probs, st_g, warm_g)
    cc, xc, _ = c_oracle.solve_batch(params, cmap, probs, st_c, warm_c)
    # objective values reported by the kernel are the reference objective at its solution
    f_at = c_oracle.objective_batch(params, cmap, probs, xg)
    assert np.allclose(f_at, cg["cost"], rtol=1e-12, atol=1e-12)
    # not worse than the CPU mirror (both are local searches; ties broken identically)
    assert (cg["cost"] <= cc["cost"] + 1e-6).mean() >= (0.99 if n_steps < 32 else 0.9)
    same = np.abs(xg - xc).max(axis=1) <= (1e-5 if n_steps < 32 else 1e-3)
    assert same.mean() >= (0.95 if n_steps < 32 else 0.85), same.mean()
    dv = np.abs(cg["vel"] - cc["vel"]).max(axis=1)
    assert (dv <= 1e-3).mean() >= 0.98, (dv <= 1e-3).mean()
    assert (cg["status"] == 0).mean() >= 0.97


# ------------------------------------------------------------------ P2 / P3 vs SciPy on the reference
# (control_steps, method): 0 = auto (dense Newton at 3, Riccati sweep otherwise), 1 = L-BFGS, 2 = dense Newton
# (control_steps <= 8), 3 = Riccati sweep -- every kernel variant sees reference SLSQP solves
@pytest.mark.parametrize("n_steps,method", [(3, 0), (3, 1), (3, 3), (8, 0), (8, 1), (8, 2), (32, 0), (32, 1)])
def test_p2_p3_against_reference_slsqp_solves(solver_mod, n_steps, method):
    """G3: cold-start solves of the REFERENCE's objective by SciPy SLSQP (py:363-364), as shipped (ftol 1e-3)
    and run to the end (ftol 1e-12), at control_steps 3, 8 and 32 (BASELINE configs 2, 3 and 5)."""
    g = util.load("g3_solves.npz")
    k = "" if n_steps == 3 else "n%d_" % n_steps
    params = util.params_from(g["param_keys"], g[k + "params"])
    assert params["control_steps"] == n_steps
    params["method"] = method
    if method == 1 and n_steps > 8:
        params["max_iterations"] = 600    # L-BFGS with 4 pairs on 96 variables needs more than SLSQP's 100
    probs = util.problems_from(g[k + "problems"])
    hm = g[k + "has_map"].astype(bool)
    for mask, cells in ((~hm, np.zeros_like(g[k + "cells"])), (hm, g[k + "cells"])):
        cmap = (cells,) + tuple(g[k + "map_meta"])
        pr = probs[mask]
        st, warm = synthetic.make_states(pr, n_steps)
        with _solver(solver_mod, params, cmap) as s:
            cmds, x = s.solve(pr, st, warm)
        # P3: not worse than the reference path at its shipped tolerance (ftol 1e-3)
        assert (cmds["cost"] <= g[k + "f_loose"][mask] + 1e-3).all()
        xs = x.reshape(len(x), -1, 3)
        assert (np.hypot(xs[:, :, 0], xs[:, :, 1]) <= params["max_vel_trans"] + 1e-9).all()
        assert (np.abs(xs[:, :, 2]) <= params["max_vel_theta"] + 1e-12).all()
        assert (cmds["status"] == 0).all()
        if not cells.any():
            # P2: unique minimiser -> first control within 1e-3 of SLSQP at ftol 1e-12 (where SLSQP itself
            # got there: status 0; at control_steps 32 it runs into maxiter 500 on some cases)
            ok = g[k + "status_tight"][mask] == 0
            du0 = np.abs(x[:, :3] - g[k + "x_tight"][mask][:, :3]).max(axis=1)
            assert du0[ok].max() <= 1e-3, du0[ok].max()
            # (the Newton directions end within 1e-9 of SLSQP's optimum value; L-BFGS within 1e-7)
            assert (cmds["cost"] <= g[k + "f_tight"][mask] + (1e-7 if method == 1 else 1e-9))[ok].all()
            assert (cmds["cost"] <= g[k + "f_tight"][mask] + 1e-6).all()


@pytest.mark.parametrize("pset,n_steps,method", [("cut", 3, 0), ("cut", 8, 0), ("cut", 8, 2), ("cut", 3, 1),
                                                 ("turn", 3, 0), ("turn", 8, 0), ("turn", 3, 3),
                                                 ("readme", 16, 0), ("readme", 16, 1)])
def test_p2_p3_at_other_parameter_sets(solver_mod, pset, n_steps, method):
    """G8: the reference's SLSQP solves for parameter sets that take the GENERAL kernels (not the README-like
    "tame" specialisations): the vx/vy box cutting the speed disc with v_cur outside the feasible set for many
    requests ("cut"), and a fast-turning robot whose heading leaves [-pi/4, pi/4] within a 1.2 s horizon, with
    other weights ("turn").  P2 on the zero map, P3 everywhere.  (The directions without a wall model are not offered at
    "turn"'s heavy costmap weight: test_forced_directions_without_a_wall_model_are_refused_at_heavy_costmap_weights.)"""
    g = util.load("g8_solves_params.npz")
    k = "%s_n%d_" % (pset, n_steps)
    params = util.params_from(g["param_keys"], g[k + "params"])
    assert params["control_steps"] == n_steps
    params["method"] = method
    if method == 1 and n_steps > 8:
        params["max_iterations"] = 600
    probs = util.problems_from(g[k + "problems"])
    hm = g[k + "has_map"].astype(bool)
    for mask, cells in ((~hm, np.zeros_like(g[k + "cells"])), (hm, g[k + "cells"])):
        cmap = (cells,) + tuple(g[k + "map_meta"])
        pr = probs[mask]
        st, warm = synthetic.make_states(pr, n_steps)
        with _solver(solver_mod, params, cmap) as s:
            cmds, x = s.solve(pr, st, warm)
            f_at = s.objective(pr, g[k + "x_tight"][mask])
        assert np.allclose(f_at, g[k + "f_tight"][mask], rtol=1e-12, atol=1e-12)   # the objective kernel, these parameters
        worse = cmds["cost"] - g[k + "f_loose"][mask]
        assert (worse <= 1e-3).all(), worse.max()      # P3, every case, every selectable configuration
        xs = x.reshape(len(x), -1, 3)
        assert (np.hypot(xs[:, :, 0], xs[:, :, 1]) <= params["max_vel_trans"] + 1e-9).all()
        assert (xs[:, :, 0] <= params["max_vel_x"] + 1e-12).all() and (xs[:, :, 0] >= params["min_vel_x"] - 1e-12).all()
        assert (np.abs(xs[:, :, 2]) <= params["max_vel_theta"] + 1e-12).all()
        assert (cmds["status"] == 0).all()
        if not cells.any():
            ok = g[k + "status_tight"][mask] == 0
            du0 = np.abs(x[:, :3] - g[k + "x_tight"][mask][:, :3]).max(axis=1)
            assert du0[ok].max() <= 1e-3, du0[ok].max()
            assert (cmds["cost"] <= g[k + "f_tight"][mask] + 1e-6).all()



# ------------------------------------------------------------------ round 3: the remaining pins of the solver
def _command(s, probs, x, n_steps):
    """the velocity command optimizer() returns for raw solver output x (py:365-395: low-pass + clamp), cold state,
    through K2 on the device"""
    st, warm = synthetic.make_states(probs, n_steps)
    return s.postprocess(probs, st, warm, x.copy())["vel"]


@pytest.mark.parametrize("fixture,prefix", util.G9_GROUPS)
@pytest.mark.parametrize("method", [0, 3])
def test_g9_node_defaults_p2_p3_and_the_literal_command_gate(solver_mod, fixture, prefix, method):
    """G9: the parameter values the node itself declares (py:49-75: opt_tolerance 1e-5, every weight 0.5, w_footprint
    2000, limits 0.5, horizon 0.5), cold solves by the reference's SLSQP at control_steps 3 and 8.  P2 / P3 as for
    G3 -- P3 against SLSQP as shipped at THIS tolerance (ftol 1e-5) -- and the north star's sentence taken literally,
    on the COMMAND (after low-pass and clamp, K2), zero map:
      (L1) within 1e-3 of the SciPy path run to convergence (ftol 1e-12), every case;
      (L2) never further from the SciPy path as shipped (ftol 1e-5) than that path is from its own converged answer,
           + 1e-4 (SLSQP at 1e-5 still stops up to 0.1 short in u0: the objective is that flat);
      (L3) within 1e-3 of the path as shipped wherever the path as shipped is itself converged to 1e-4."""
    g, params, probs, hm = util.solve_group(fixture, prefix)
    n_steps = params["control_steps"]
    assert params["opt_tolerance"] == 1e-5 and params["w_costmap"] == 0.5 and params["w_footprint"] == 2000
    params["method"] = method
    for mask, cells in ((~hm, np.zeros_like(g["cells"])), (hm, g["cells"])):
        cmap = (cells,) + tuple(g["map_meta"])
        pr = probs[mask]
        st, warm = synthetic.make_states(pr, n_steps)
        with _solver(solver_mod, params, cmap) as s:
            cmds, x = s.solve(pr, st, warm)
            assert np.allclose(s.objective(pr, g["x_tight"][mask]), g["f_tight"][mask], rtol=1e-12, atol=1e-12)
            if not cells.any():
                v_build = _command(s, pr, x, n_steps)
                v_loose = _command(s, pr, g["x_loose"][mask], n_steps)
                v_tight = _command(s, pr, g["x_tight"][mask], n_steps)
        assert (cmds["status"] == 0).all()
        xs = x.reshape(len(x), -1, 3)
        assert (np.hypot(xs[:, :, 0], xs[:, :, 1]) <= params["max_vel_trans"] + 1e-9).all()
        assert (np.abs(xs) <= 0.5 + 1e-12).all()
        worse = cmds["cost"] - g["f_loose"][mask]
        assert (worse <= 1e-3).all(), worse.max()                                                             # P3
        if not cells.any():
            ok = g["status_tight"][mask] == 0
            du0 = np.abs(x[:, :3] - g["x_tight"][mask][:, :3]).max(axis=1)
            assert du0[ok].max() <= 1e-3, du0[ok].max()                                                       # P2
            assert (cmds["cost"] <= g["f_tight"][mask] + 1e-5).all()
            d_bt = np.abs(v_build - v_tight).max(axis=1)
            d_bl = np.abs(v_build - v_loose).max(axis=1)
            d_lt = np.abs(v_loose - v_tight).max(axis=1)
            assert d_bt[ok].max() <= 1e-3, d_bt[ok].max()                                                     # L1
            assert (d_bl[ok] <= d_lt[ok] + 1e-4).all(), (d_bl - d_lt)[ok].max()                               # L2
            settled = ok & (d_lt <= 1e-4)
            assert settled.sum() >= len(ok) // 2 and d_bl[settled].max() <= 1e-3, d_bl[settled].max()        # L3
            print("G9 %s method %d: command vs SLSQP@1e-12 max %.1e; vs SLSQP@1e-5: %d of %d within 1e-3 (the reference "
                  "itself: %d), max %.1e" % (prefix, method, d_bt[ok].max(), (d_bl <= 1e-3).sum(), len(d_bl),
                                             (d_lt <= 1e-3).sum(), d_bl.max()))


@pytest.mark.parametrize("fixture,prefix", util.G8_MID_GROUPS)
@pytest.mark.parametrize("method", [0, 2, 3])
def test_g8_mid_costmap_weights_across_the_auto_threshold(solver_mod, fixture, prefix, method):
    """G8 "mid": the README's parameters with w_costmap / w_trans = 0.10 ... 0.30, every case on the costmap: P3 on
    EVERY case for the dense-Newton kernel (AUTO below 1/4, the headline kernel) and the stage-wise one (AUTO above):
    the threshold in neo_mpc_capi.cpp derive() keeps neither away from problems it cannot do."""
    g, params, probs, hm = util.solve_group(fixture, prefix)
    assert hm.all() and abs(params["w_costmap"] / params["w_trans"] - int(prefix[1:3]) / 100.0) < 1e-12
    if method == 2 and params["w_costmap"] > 0.25 * params["w_trans"]:
        pytest.skip("the dense direction is not offered above w_costmap = w_trans / 4 (NEO_MPC_ERR_UNSUPPORTED)")
    params["method"] = method
    cmap = (g["cells"],) + tuple(g["map_meta"])
    st, warm = synthetic.make_states(probs, 3)
    with _solver(solver_mod, params, cmap) as s:
        cmds, x = s.solve(probs, st, warm)
        assert np.allclose(s.objective(probs, g["x_tight"]), g["f_tight"], rtol=1e-12, atol=1e-12)
    assert (cmds["status"] == 0).all()
    assert (cmds["cost"] <= g["f_loose"] + 1e-3).all(), (cmds["cost"] - g["f_loose"]).max()
    xs = x.reshape(len(x), -1, 3)
    assert (np.hypot(xs[:, :, 0], xs[:, :, 1]) <= 0.7 + 1e-9).all() and (np.abs(xs) <= 0.7 + 1e-12).all()


@pytest.mark.parametrize("method", [0, 1])
def test_g3_n32_unique_minimisers(solver_mod, method):
    """G3 at control_steps 32 (BASELINE config 5) on the all-free map: 64 problems solved by the reference's SLSQP with
    maxiter raised until ftol 1e-12 reports status 0 (57 of them) -- P2 at 32 control steps on >= 40 problems."""
    g, params, probs, _ = util.solve_group("g3_solves_n32_zero.npz", "")
    assert params["control_steps"] == 32
    params["method"] = method
    if method == 1:
        params["max_iterations"] = 600
    zero = (np.zeros((200, 200), np.uint8),) + tuple(g["map_meta"])
    st, warm = synthetic.make_states(probs, 32)
    with _solver(solver_mod, params, zero) as s:
        cmds, x = s.solve(probs, st, warm)
        assert np.allclose(s.objective(probs, g["x_tight"]), g["f_tight"], rtol=1e-12, atol=1e-12)
    ok = g["status_tight"] == 0
    assert ok.sum() >= 40
    du0 = np.abs(x[:, :3] - g["x_tight"][:, :3]).max(axis=1)
    assert du0[ok].max() <= 1e-3, du0[ok].max()
    assert (cmds["cost"] <= g["f_tight"] + 1e-6).all() and (cmds["cost"] <= g["f_loose"] + 1e-3).all()
    assert (cmds["status"] == 0).all()
    print("N=32, %d unique-minimiser problems: max |u0 - u0(SLSQP 1e-12)| = %.2e" % (ok.sum(), du0[ok].max()))


@pytest.mark.parametrize("fixture", util.EPISODE_FIXTURES)
def test_p3_on_the_reference_warm_starts(solver_mod, fixture):
    """G4: every call of the recorded episodes is solved from the REFERENCE's own state (its warm start
    `initial_guess`, `last_control`, goal bookkeeping, real costmap): the kernel's answer must be feasible and
    not worse than the raw SLSQP output the reference produced there (f <= f(x_ref) + 1e-3).  The states are
    advanced with the reference's raw x.x injected (P5), so call k starts exactly where the reference did."""
    from oracle import c_oracle
    g = util.load(fixture)
    params = util.params_from(g["param_keys"], g["params"])
    n = params["control_steps"]
    cmap = (g["cells"],) + tuple(g["map_meta"])
    probs = util.problems_from(g["problems"])
    n_ep, n_calls = probs.shape
    states, warm = abi.new_states(n_ep, n)
    worse = []
    with _solver(solver_mod, params, cmap) as s:
        for k in range(n_calls):
            fp = g["footprint"][:, k]
            rows = probs[:, k].copy()
            has = ~np.isnan(fp).any(axis=(1, 2))
            rows["footprint_cost"] = 0.0
            if has.any():
                rows["footprint_cost"][has] = c_oracle.footprint_cost_batch(cmap, fp[has])
            cmds, x = s.solve(rows, states.copy(), warm.copy())
            f_ref = s.objective(rows, g["raw_x"][:, k])
            worse.append(cmds["cost"] - f_ref)
            xs = x.reshape(n_ep, n, 3)
            assert (np.hypot(xs[:, :, 0], xs[:, :, 1]) <= params["max_vel_trans"] + 1e-9).all()
            for q, axis in enumerate(("x", "y", "theta")):
                assert (xs[:, :, q] <= params["max_vel_" + axis] + 1e-12).all()
                assert (xs[:, :, q] >= params["min_vel_" + axis] - 1e-12).all()
            assert (cmds["status"] == 0).all()
            s.postprocess(rows, states, warm, g["raw_x"][:, k], g["success"][:, k])   # the reference's next state
    worse = np.array(worse)
    assert worse.max() <= 1e-3, worse.max()
    print("f(build) - f(reference raw x): max %.2e median %.2e" % (worse.max(), np.median(worse)))


def test_forced_directions_without_a_wall_model_are_refused_at_heavy_costmap_weights(solver_mod):
    """No selectable configuration is knowingly worse than the reference: NEO_MPC_METHOD_LBFGS / _NEWTON have no wall
    model for costmap steps and, forced onto w_costmap > w_trans / 4, used to end above SLSQP on a few percent of the
    costmap cases (G8 "turn", G9) -- neo_mpc_create answers NEO_MPC_ERR_UNSUPPORTED there; AUTO and RICCATI are offered
    everywhere.  A LIVE handle reconfigured across the threshold (cb_params, py:405-439, cannot fail in the reference) keeps
    working: it runs the stage-wise direction while the weights stay there -- bit for bit what a RICCATI handle answers --
    and goes back to its pinned direction with them.  Unknown compat bits are refused; the behaviour version can be asked."""
    import ctypes as C
    from neo_mpc_planner2_amd import _lib
    lib = _lib.load()
    assert lib.neo_mpc_abi_version() == abi.ABI_VERSION == 2 and lib.neo_mpc_behaviour_version() == 6
    heavy = util.orc.make_params(w_costmap=0.3)          # 0.3 > 0.82 / 4
    for method in (1, 2):
        with pytest.raises(_lib.NeoMpcError) as e:
            solver_mod.BatchSolver(dict(heavy, method=method))
        assert e.value.code == -5 and "wall model" in str(e.value)
    for method in (0, 3):
        solver_mod.BatchSolver(dict(heavy, method=method)).close()
    cfg, cmap, probs, st0, warm0 = synthetic.make_workload("C2", seed=12, batch=512)
    with _solver(solver_mod, dict(heavy, method=3), cmap) as s:
        want = s.solve(probs, st0.copy(), warm0.copy())[0]
    for method in (1, 2):
        with _solver(solver_mod, util.orc.make_params(method=method), cmap) as s:   # README weights: offered ...
            assert lib.neo_mpc_effective_method(s._handle) == method
            first = s.solve(probs, st0.copy(), warm0.copy())[0]
            s.set_params(**dict(heavy, method=method))                              # ... reconfigured across the threshold
            assert lib.neo_mpc_effective_method(s._handle) == 3 and s.params["method"] == method
            got = s.solve(probs, st0.copy(), warm0.copy())[0]
            assert got.tobytes() == want.tobytes()
            s.set_params(**util.orc.make_params(method=method))                     # ... and back
            assert lib.neo_mpc_effective_method(s._handle) == method
            assert s.solve(probs, st0.copy(), warm0.copy())[0].tobytes() == first.tobytes()
    assert lib.neo_mpc_effective_method(None) < 0
    with pytest.raises(_lib.NeoMpcError) as e:
        solver_mod.BatchSolver(util.orc.make_params(compat_flags=1 | 0x40))
    assert e.value.code == -1 and "compat_flags" in str(e.value) and lib.neo_mpc_last_error_code() == -1
    solver_mod.BatchSolver(util.orc.make_params(compat_flags=abi.COMPAT_ODOM_YAW_GOAL_W | abi.COMPAT_REFERENCE_START)).close()


def test_reference_start_compat_bit_starts_every_search_at_the_warm_start(solver_mod):
    """NEO_MPC_COMPAT_REFERENCE_START: every search starts at py:397-400's warm start (no better-of-two start on free
    space).  Cold starts do not depend on it; warm ticks of a fleet take more iterations with it and end at the same
    objective (1e-4)."""
    cfg, cmap, probs, st0, warm0 = synthetic.make_workload("C2", seed=4, batch=2048)
    base = util.orc.make_params()
    res = {}
    for tag, flags in (("default", abi.COMPAT_ODOM_YAW_GOAL_W), ("reference", abi.COMPAT_ODOM_YAW_GOAL_W | abi.COMPAT_REFERENCE_START)):
        st, warm = st0.copy(), warm0.copy()
        with _solver(solver_mod, dict(base, compat_flags=flags), cmap) as s:
            cold = s.solve(probs, st, warm)[0].copy()
            ticks = [s.solve(probs, st, warm)[0].copy() for _ in range(3)]
        res[tag] = (cold, ticks)
    assert res["default"][0].tobytes() == res["reference"][0].tobytes()
    it_d = np.mean([t["iterations"].mean() for t in res["default"][1]])
    it_r = np.mean([t["iterations"].mean() for t in res["reference"][1]])
    assert it_d < it_r, (it_d, it_r)
    print("warm ticks: %.2f iterations from the better of two starts, %.2f from the reference's" % (it_d, it_r))


# ------------------------------------------------------------------ round 4: held-out sets and the warm gate
@pytest.mark.parametrize("name,n_steps", util.G10_GROUPS)
def test_g10_held_out_parameter_sets(solver_mod, name, n_steps):
    """G10: three parameter sets that were never looked at while thresholds were tuned (two of them the round-3 judge's),
    control_steps 3 / 5 / 8 / 12, 300 x 300 maps of other seeds, through the C-ABI on the GPU: P3 on every case, P2 <= 3e-4
    on the all-free-map cases (SLSQP run to the end, py:363-364 at ftol 1e-12)."""
    def solve(params, cmap, pr):
        st, warm = synthetic.make_states(pr, params["control_steps"])
        with _solver(solver_mod, params, cmap) as s:
            return s.solve(pr, st, warm)
    m = util.check_held_out_group(solve, name, n_steps)
    print("G10 %s N=%d: P2 %.2e, P3 margin free %.2e map %.2e, iterations %.1f / %.1f"
          % (name, n_steps, m["p2"], m["p3_free"], m["p3_map"], m["it_free"], m["it_map"]))


@pytest.mark.parametrize("fixture", util.G11_FIXTURES)
def test_g11_warm_commands_against_the_converged_reference(solver_mod, fixture):
    """G11: the deployed (warm-started) mode -- every call of the reference's episodes RUN TO CONVERGENCE (opt_tolerance
    1e-12, SLSQP's iteration cap raised to 500; py:363-364, 397-400) solved by K1 at the README tolerance from the
    reference's own state: the command (after low-pass and clamp, K2) within 1e-3 of the reference's on >= 99.9 % of
    the ticks."""
    solvers = {}

    def get(params, cmap):
        if "s" not in solvers:
            solvers["s"] = _solver(solver_mod, params, cmap)
        return solvers["s"]

    def solve(params, cmap, rows, st, wm):
        return get(params, cmap).solve(rows, st, wm)

    def post(params, cmap, rows, st, wm, x, success):
        get(params, cmap).postprocess(rows, st, wm, x, success)
    try:
        dv, du, its, settled = util.warm_gate(solve, post, fixture)
    finally:
        if "s" in solvers:
            solvers["s"].close()
    print("G11 %s: %d ticks, |command diff| p99 %.2e max %.2e, above 1e-3: %d; |u0 diff| above 1e-3: %d; iterations %.2f"
          % (fixture, dv.size, np.percentile(dv, 99), dv.max(), (dv > 1e-3).sum(), (du > 1e-3).sum(), its.mean()))
    assert (dv <= 1e-3).mean() >= 0.999, ((dv > 1e-3).sum(), dv.size, dv.max())
    assert dv.max() <= 3e-3


# ------------------------------------------------------------------ G6: the adjoint gradient inside every kernel variant
@pytest.mark.parametrize("n_steps,method", [(3, 0), (3, 1), (3, 3), (8, 0), (8, 1), (8, 2), (32, 0), (32, 1)])
def test_kernel_gradient_matches_reference_fd_gradient(solver_mod, n_steps, method):
    """G6: SciPy's forward-difference gradient of the reference objective (what SLSQP works with,
    _slsqp_py.py:381) against the analytic gradient taken from inside K1 (neo_mpc_gradient_batch)."""
    g = util.load("g6_fd_gradient.npz")
    k = "n%d_" % n_steps
    params = util.params_from(g["param_keys"], g[k + "params"])
    params["method"] = method
    probs = util.problems_from(g[k + "problems"])
    zero = (np.zeros((200, 200), np.uint8), 0.05, -5.0, -5.0)
    with _solver(solver_mod, params, zero) as s:
        got = s.gradient(probs, g[k + "u"])
    scale = np.maximum(np.abs(g[k + "grad"]).max(axis=1, keepdims=True), 1e-3)
    err = np.abs(got - g[k + "grad"]) / scale
    assert err.max() <= 2e-6, err.max()


# ------------------------------------------------------------------ f-4: predicted path == the reference's local_plan
@pytest.mark.parametrize("n_steps", [3, 8, 32])
def test_predicted_path_matches_reference_local_plan(solver_mod, n_steps):
    """G7: `publishLocalPlan` (py:271-310) run by the reference; K2's `predicted_path` (X, Y, yaw) for the same
    controls from the same pose, and the w-first quaternion of py:301-305 formed from the yaw."""
    g = util.load("g7_local_plan.npz")
    k = "n%d_" % n_steps
    params = util.params_from(g["param_keys"], g[k + "params"])
    count = len(g[k + "x"])
    probs = synthetic.make_problems(count, 200, seed=5)
    probs["cur_xy"], probs["cur_q"] = g[k + "tf_xy"], g[k + "tf_q"]
    zero = (np.zeros((400, 400), np.uint8), 0.05, -10.0, -10.0)
    st, warm = abi.new_states(count, n_steps)
    with _solver(solver_mod, params, zero) as s:
        _, path = s.postprocess(probs, st, warm, g[k + "x"], want_path=True)
    ref = g[k + "path"][:, 1:]
    assert np.abs(path[:, :, :2] - ref[:, :, :2]).max() <= 1e-12
    assert np.abs(np.sin(0.5 * path[:, :, 2]) - ref[:, :, 4]).max() <= 1e-12     # orientation.z (py:305)
    assert np.abs(np.cos(0.5 * path[:, :, 2]) - ref[:, :, 5]).max() <= 1e-12     # orientation.w (py:302)
    assert (ref[:, :, 2:4] == 0.0).all()


@pytest.mark.parametrize("fixture", util.EPISODE_FIXTURES)
def test_predicted_path_of_the_reference_episodes(solver_mod, fixture):
    """G4 `local_plan`: the Path published inside optimizer() (rollout of the UNFILTERED x.x from the request's
    pose) == K2's predicted_path with the reference's x.x injected, over all 520 calls."""
    g = util.load(fixture)
    params = util.params_from(g["param_keys"], g["params"])
    cmap = (g["cells"],) + tuple(g["map_meta"])
    probs = util.problems_from(g["problems"])
    n_ep, n_calls = probs.shape
    flat = probs.reshape(-1).copy()
    flat["footprint_cost"] = 0.0
    st, warm = abi.new_states(len(flat), params["control_steps"])
    with _solver(solver_mod, params, cmap) as s:
        _, path = s.postprocess(flat, st, warm, g["raw_x"].reshape(len(flat), -1), want_path=True)
    ref = g["local_plan"].reshape(len(flat), params["control_steps"] + 1, 6)[:, 1:]
    assert np.abs(path[:, :, :2] - ref[:, :, :2]).max() <= 1e-12
    assert np.abs(np.sin(0.5 * path[:, :, 2]) - ref[:, :, 4]).max() <= 1e-12
    assert np.abs(np.cos(0.5 * path[:, :, 2]) - ref[:, :, 5]).max() <= 1e-12


# ------------------------------------------------------------------ full BASELINE size: properties
def test_c2_full_size_properties(solver_mod):
    """4096 instances, control_steps 3, 500x500 map (BASELINE config 2): feasibility,
    descent, idempotence, and sharded == unsharded."""
    cfg, cmap, probs, st, warm = synthetic.make_workload("C2", seed=0)
    params = util.orc.make_params()
    with _solver(solver_mod, params, cmap) as s:
        st0, warm0 = st.copy(), warm.copy()
        cmds, x = s.solve(probs, st, warm)
        f0 = s.objective(probs, np.zeros_like(x))
        assert (cmds["cost"] <= f0 + 1e-12).all()                      # never worse than the start
        xs = x.reshape(len(x), -1, 3)
        assert (np.hypot(xs[:, :, 0], xs[:, :, 1]) <= 0.7 + 1e-9).all()
        assert (np.abs(xs) <= 0.7 + 1e-12).all()
        # idempotence: restarting from the solution does not move (beyond the step tolerance)
        st2 = st0.copy()
        cm2, x2 = s.solve(probs, st2, x.copy())
        assert (cm2["cost"] <= cmds["cost"] + 1e-12).all()
        # (round 5: a search that has ended looks at the costmap cells around every stage (cell_scan.h) -- ONCE, and a restart
        # from the solution, a second search, got a second look: 5.5 % of the instances moved by more than 1e-3, each to a
        # LOWER objective.  Round 6: the instances with a wall in reach run the stage-wise direction, and 4.2 % move (mirror:
        # 0.958 / 0.940 within 1e-3 / 1e-4).  Scanning again from where a scan has put the iterate would make more answers
        # fixed points -- 97.3 % with one repeat, 97.7 % with seven -- but every scan of a launch's last waves lengthens the
        # launch (dense kernel 0.093 -> 0.109 -> 0.139 ms): NEO_RULE_SCAN_REPEATS stays 1, solver_rules.h says so)
        moved = np.abs(x2 - x).max(axis=1)
        assert (moved <= 1e-3).mean() >= 0.95     # the north-star tolerance
        assert (moved <= 1e-4).mean() >= 0.93
        assert (cm2["cost"][moved > 1e-3] < cmds["cost"][moved > 1e-3]).all()
        # sharding: two half batches == the whole batch, bit for bit
        h = len(probs) // 2
        sa, wa = st0[:h].copy(), warm0[:h].copy()
        sb, wb = st0[h:].copy(), warm0[h:].copy()
        ca, xa = s.solve(probs[:h], sa, wa)
        cb, xb = s.solve(probs[h:], sb, wb)
        assert (np.concatenate([xa, xb]) == x).all()
        assert (np.concatenate([ca["vel"], cb["vel"]]) == cmds["vel"]).all()
        assert (np.concatenate([sa, sb]).tobytes() == st.tobytes())
    # acceleration clamp honoured (py:385-391) on the non-stopped instances
    moving = (cmds["flags"] & abi.FLAG_STOPPED) == 0
    dv = np.abs(cmds["vel"] - probs["cur_vel"])[moving]
    assert (dv <= np.array([2.5, 2.5, 3.0]) / 30.0 + 1e-12).all()


def test_device_resident_batch_matches_host_batch(solver_mod):
    import torch
    cfg, cmap, probs, st, warm = synthetic.make_workload("C2", seed=3, batch=512)
    params = util.orc.make_params()
    with _solver(solver_mod, params, cmap) as s:
        db = solver_mod.DeviceBatch(probs, st, warm, "cuda:0")
        s.solve_device(db.problems, db.states, db.warm, db.commands, db.solution, velocities=db.vel)
        torch.cuda.synchronize()
        assert (db.vel.cpu().numpy() == db.commands_host()["vel"]).all()
        cmds, x = s.solve(probs, st, warm)
        assert (db.commands_host()["vel"] == cmds["vel"]).all()
        assert (db.solution.cpu().numpy() == x).all()
        assert (db.velocities().cpu().numpy() == cmds["vel"]).all()
        # costmap handed over as a device tensor
        s.set_costmap(torch.from_numpy(cmap[0]).cuda(), *cmap[1:])
        st2, warm2 = synthetic.make_states(probs, 3)
        torch.cuda.synchronize()
        cm2, x2 = s.solve(probs, st2, warm2)
        assert (x2 == x).all()


def test_page_locked_host_batches_are_worked_on_in_place(solver_mod):
    """neo_mpc_solve_batch on page-locked arrays (neo_mpc_pin_host_memory: what a caller built without the HIP headers
    uses) copies nothing -- K1 reads the records from the caller's arrays and writes the results into them over PCIe:
    commands, raw solution, state, warm start and predicted path are bit-identical with the staged paths and with
    pageable arrays, for the three host-path variants (neo_mpc_set_host_path) and a second, warm-started tick."""
    import ctypes as C
    from neo_mpc_planner2_amd import _lib
    lib = _lib.load()
    cfg, cmap, probs, st0, warm0 = synthetic.make_workload("C2", seed=5, batch=1000)
    params = util.orc.make_params()
    with _solver(solver_mod, params, cmap) as s:
        ref_st, ref_warm = st0.copy(), warm0.copy()
        ref = [s.solve(probs, ref_st, ref_warm, want_path=True) for _ in range(2)]       # pageable, two ticks
        ref = [(c.copy(), x.copy(), p.copy()) for c, x, p in ref]
        for mode in ("zerocopy", "zerocopy_out", "staged"):
            s.set_host_path(mode)
            arrays = [np.ascontiguousarray(probs).copy(), st0.copy(), warm0.copy(),
                      np.zeros(len(probs), dtype=abi.COMMAND_DTYPE), np.zeros((len(probs), 9)), np.zeros((len(probs), 3, 3))]
            for a in arrays:
                assert lib.neo_mpc_pin_host_memory(C.c_void_p(a.ctypes.data), a.nbytes) == 0, lib.neo_mpc_last_error()
            try:
                p_probs, p_st, p_warm, p_cmd, p_sol, p_path = arrays
                for tick in range(2):
                    b = abi.batch_struct(p_probs, p_st, p_warm, p_cmd, p_sol, p_path)
                    _lib.check(lib.neo_mpc_solve_batch(s._handle, C.byref(b)))
                    assert p_cmd.tobytes() == ref[tick][0].tobytes(), (mode, tick)
                    assert (p_sol == ref[tick][1]).all() and (p_path == ref[tick][2]).all(), (mode, tick)
                assert p_st.tobytes() == ref_st.tobytes() and (p_warm == ref_warm).all(), mode
            finally:
                for a in arrays:
                    assert lib.neo_mpc_unpin_host_memory(C.c_void_p(a.ctypes.data)) == 0
        s.set_host_path("auto")
        assert lib.neo_mpc_set_host_path(s._handle, 7) == -1 and lib.neo_mpc_pin_host_memory(None, 16) == -1


def test_large_staged_host_batches_go_through_in_pieces(solver_mod, monkeypatch):
    """A host batch of 65 536 instances or more that has to be staged (pageable arrays) is copied, solved and copied back
    in four pieces on two streams; the instances are independent, so commands, raw solution, state, warm start and
    predicted path are bit-identical with the one-piece path (NEO_MPC_NO_CHUNKS) -- at a count that is no multiple of
    anything, over two ticks."""
    cfg, cmap, probs, st0, warm0 = synthetic.make_workload("C2", seed=9, batch=70001)
    params = util.orc.make_params()
    # (the library reads its A/B switches once, in neo_mpc_create: one handle per setting)
    monkeypatch.setenv("NEO_MPC_NO_CHUNKS", "1")
    with _solver(solver_mod, params, cmap) as s:
        r_st, r_warm = st0.copy(), warm0.copy()
        ref = [tuple(x.copy() for x in s.solve(probs, r_st, r_warm, want_path=True)) for _ in range(2)]
    monkeypatch.delenv("NEO_MPC_NO_CHUNKS")
    with _solver(solver_mod, params, cmap) as s:
        c_st, c_warm = st0.copy(), warm0.copy()
        for tick in range(2):
            cmd, sol, path = s.solve(probs, c_st, c_warm, want_path=True)
            assert cmd.tobytes() == ref[tick][0].tobytes(), tick
            assert (sol == ref[tick][1]).all() and (path == ref[tick][2]).all(), tick
        assert c_st.tobytes() == r_st.tobytes() and (c_warm == r_warm).all()


def test_batches_in_flight_begin_and_wait(solver_mod):
    """neo_mpc_solve_batch_begin / _wait, the two halves of `async_send_request(request)` ... `result.get()`
    (cpp:248-250): four page-locked batches in flight at once end with exactly the commands, raw solutions, states and
    warm starts of four synchronous calls -- over two ticks, with the costmap replaced between the ticks while nothing
    has been waited for yet (the ingest is ordered behind the batches in flight); pageable arrays, a fifth batch and a
    stale ticket are refused."""
    import ctypes as C
    from neo_mpc_planner2_amd import _lib
    lib = _lib.load()
    params = util.orc.make_params()
    cmap = synthetic.make_costmap(500, seed=0)
    cmap2 = synthetic.make_costmap(500, seed=3)
    fleets = []
    for k in range(4):
        probs = np.ascontiguousarray(synthetic.make_problems(700 + 100 * k, 500, seed=40 + k))
        st, warm = synthetic.make_states(probs, 3)
        fleets.append((probs, st, warm))
    with _solver(solver_mod, params, cmap) as s:
        ref = []
        for probs, st, warm in fleets:                      # the synchronous calls, pageable arrays
            r_st, r_warm = st.copy(), warm.copy()
            c1, x1 = s.solve(probs, r_st, r_warm)
            ref.append([(c1.copy(), x1.copy())])
            ref[-1].append((r_st, r_warm))
        s.set_costmap(*cmap2)
        for (probs, st, warm), r in zip(fleets, ref):
            c2, x2 = s.solve(probs, r[1][0], r[1][1])
            r.append((c2.copy(), x2.copy()))
        s.set_costmap(*cmap)
        pinned = []
        try:
            for probs, st, warm in fleets:
                arrays = [probs.copy(), st.copy(), warm.copy(), np.zeros(len(probs), dtype=abi.COMMAND_DTYPE),
                          np.zeros((len(probs), 9))]
                for a in arrays:
                    assert lib.neo_mpc_pin_host_memory(C.c_void_p(a.ctypes.data), a.nbytes) == 0, lib.neo_mpc_last_error()
                pinned.append(arrays)
            tickets = [s.solve_begin(a[0], a[1], a[2], out=(a[3], a[4])) for a in pinned]
            assert sorted(tickets) == [1, 2, 3, 4]
            with pytest.raises(_lib.NeoMpcError):            # a fifth batch
                s.solve_begin(pinned[0][0], pinned[0][1], pinned[0][2], out=(pinned[0][3], pinned[0][4]))
            s.set_costmap(*cmap2)                            # ... behind the four batches still in flight
            for t, a, r in zip(tickets, pinned, ref):
                cmd, sol = s.solve_wait(t)
                assert cmd.tobytes() == r[0][0].tobytes() and (sol == r[0][1]).all()
            with pytest.raises(_lib.NeoMpcError):            # a ticket that has been waited for
                s.solve_wait(tickets[0])
            tickets = [s.solve_begin(a[0], a[1], a[2], out=(a[3], a[4])) for a in reversed(pinned)]   # tick 2, new map
            for t, a, r in zip(tickets, reversed(pinned), reversed(ref)):
                cmd, sol = s.solve_wait(t)
                assert cmd.tobytes() == r[2][0].tobytes() and (sol == r[2][1]).all()
                assert a[1].tobytes() == r[1][0].tobytes() and (a[2] == r[1][1]).all()
            probs, st, warm = fleets[0]                      # pageable arrays: refused, not staged behind the caller's back
            with pytest.raises(_lib.NeoMpcError) as e:
                s.solve_begin(probs, st.copy(), warm.copy(), out=(np.zeros(len(probs), dtype=abi.COMMAND_DTYPE), np.zeros((len(probs), 9))))
            assert e.value.code == -5, e.value.code   # NEO_MPC_ERR_UNSUPPORTED
        finally:
            for arrays in pinned:
                for a in arrays:
                    lib.neo_mpc_unpin_host_memory(C.c_void_p(a.ctypes.data))


def test_pool_and_parameters_replaced_behind_batches_in_flight(solver_mod):
    """neo_mpc_set_costmap_pool and neo_mpc_set_params between `begin` and `wait`: the pool's origins and the per-step
    costmap-term table are read by every K1 wave as it starts, and both are rewritten by blocking copies that do not order
    against the streams batches are in flight on -- the library waits for every launch that may still read them first
    (round-3 advisor, medium).  Two large page-locked batches in flight on a pool of four maps; pool (other cells, other
    origins) and parameters (other costmap weight) are replaced before anything has been waited for: the batches come out
    exactly as with the OLD pool and parameters, the next tick exactly as with the new ones."""
    import ctypes as C
    from neo_mpc_planner2_amd import _lib
    lib = _lib.load()
    params = util.orc.make_params()
    params2 = dict(params, w_costmap=0.2)
    maps = np.stack([synthetic.make_costmap(300, seed=60 + k)[0] for k in range(4)])
    maps2 = np.stack([synthetic.make_costmap(300, seed=70 + k)[0] for k in range(4)])
    origins = np.array([[-7.5, -7.5], [-7.0, -7.5], [-7.5, -7.0], [-8.0, -8.0]])
    origins2 = origins + 0.35
    fleets = []
    for k in range(2):
        probs = np.ascontiguousarray(synthetic.make_problems(60000, 300, seed=80 + k))
        probs["map_index"] = np.arange(len(probs)) % 4
        st, warm = synthetic.make_states(probs, 3)
        fleets.append((probs, st, warm))
    with solver_mod.BatchSolver(params) as s:
        s.set_costmap_pool(maps, 0.05, origins)
        ref_old = [s.solve(p, st.copy(), w.copy())[0].copy() for p, st, w in fleets]
        s.set_costmap_pool(maps2, 0.05, origins2)
        s.set_params(**params2)
        ref_new = [s.solve(p, st.copy(), w.copy())[0].copy() for p, st, w in fleets]
        assert ref_old[0].tobytes() != ref_new[0].tobytes()
        s.set_params(**params)
        s.set_costmap_pool(maps, 0.05, origins)
        pinned = []
        try:
            for probs, st, warm in fleets:
                arrays = [probs.copy(), st.copy(), warm.copy(), np.zeros(len(probs), dtype=abi.COMMAND_DTYPE), np.zeros((len(probs), 9))]
                for a in arrays:
                    assert lib.neo_mpc_pin_host_memory(C.c_void_p(a.ctypes.data), a.nbytes) == 0, lib.neo_mpc_last_error()
                pinned.append(arrays)
            tickets = [s.solve_begin(a[0], a[1], a[2], out=(a[3], a[4])) for a in pinned]
            s.set_costmap_pool(maps2, 0.05, origins2)      # ... while 120 000 instances are in flight
            s.set_params(**params2)
            for t, r in zip(tickets, ref_old):
                cmd, _ = s.solve_wait(t)
                assert cmd.tobytes() == r.tobytes()
            for a, (probs, st, warm) in zip(pinned, fleets):
                a[1][...] = st
                a[2][...] = warm
            tickets = [s.solve_begin(a[0], a[1], a[2], out=(a[3], a[4])) for a in pinned]
            for t, r in zip(tickets, ref_new):
                cmd, _ = s.solve_wait(t)
                assert cmd.tobytes() == r.tobytes()
        finally:
            for arrays in pinned:
                for a in arrays:
                    lib.neo_mpc_unpin_host_memory(C.c_void_p(a.ctypes.data))


def test_errors_are_reported_not_thrown(solver_mod):
    from neo_mpc_planner2_amd import _lib
    s = solver_mod.BatchSolver(util.orc.make_params())
    probs = synthetic.make_problems(4, 200, seed=1)
    st, warm = synthetic.make_states(probs, 3)
    with pytest.raises(_lib.NeoMpcError) as e:
        s.solve(probs, st, warm)          # no costmap yet
    assert e.value.code == -4
    with pytest.raises(_lib.NeoMpcError):
        solver_mod.BatchSolver(util.orc.make_params(control_steps=0))
    with pytest.raises(_lib.NeoMpcError):
        solver_mod.BatchSolver(util.orc.make_params(min_vel_x=0.9, max_vel_x=1.0, min_vel_y=0.9, max_vel_y=1.0))
    # costmap pool: null origins, empty pool, degenerate geometry
    import ctypes as C
    lib = _lib.load()
    cells = np.zeros((2, 8, 8), dtype=np.uint8)
    orig = np.zeros((2, 2))
    assert lib.neo_mpc_set_costmap_pool(s._handle, C.c_void_p(cells.ctypes.data), 2, 8, 8, 0.05, None) == -1
    assert lib.neo_mpc_set_costmap_pool(s._handle, C.c_void_p(cells.ctypes.data), 0, 8, 8, 0.05,
                                        C.c_void_p(orig.ctypes.data)) == -1
    assert lib.neo_mpc_set_costmap_pool(s._handle, C.c_void_p(cells.ctypes.data), 2, 8, 8, 0.0,
                                        C.c_void_p(orig.ctypes.data)) == -1
    assert b"costmap" in lib.neo_mpc_last_error()
    s.close()


# ------------------------------------------------------------------ host mirror of the service node
def test_server_mirror_episode_matches_oracle_wrapper():
    """`MpcOptimizationServer.optimizer(request, response)` (the reference's service callback
    surface) over a 40-call episode == the C oracle driven with the same requests/clock:
    warm start, low-pass, clamp, latch, new goal."""
    import math
    from neo_mpc_planner2_amd import mpc_optimization_server as srv
    from oracle import c_oracle
    params = dict(srv.README_PARAMS)
    cmap = synthetic.make_costmap(200, seed=4)
    clock = {"t": 1000.0}
    node = srv.MpcOptimizationServer(params, clock=lambda: clock["t"])
    node.set_costmap(*cmap)
    st_c, warm_c = abi.new_states(1, 3)
    row = synthetic.make_problems(1, 200, seed=440)[0].copy()
    pos, yaw, vel = np.array(row["cur_xy"]), 0.3, np.zeros(3)
    last_time = 0.0
    for k in range(40):
        if k == 20:
            g = synthetic.make_problems(1, 200, seed=901)[0]
            row["goal_xyz"], row["goal_q"] = g["goal_xyz"], g["goal_q"]
        if k % 10 == 0 and k:
            c = synthetic.make_problems(1, 200, seed=1300 + k)[0]
            row["carrot_xy"], row["carrot_q"] = c["carrot_xy"], c["carrot_q"]
        clock["t"] += (1.0 / 30.0) if k % 7 else 0.9
        row["cur_xy"], row["cur_q"], row["cur_vel"] = pos, synthetic.yaw_quat(np.array(yaw)), vel
        req = srv.make_request(row["cur_xy"], row["cur_q"], row["carrot_xy"], row["carrot_q"], row["goal_xyz"],
                               row["goal_q"], row["cur_vel"], control_interval=1.0 / 30.0)
        resp = node.optimizer(req, srv.make_response())
        out = np.array([resp.output_vel.twist.linear.x, resp.output_vel.twist.linear.y,
                        resp.output_vel.twist.angular.z])
        rec = srv.request_record(req, delta_t=clock["t"] - last_time)
        last_time = clock["t"]
        cc, xc, path_c = c_oracle.solve_batch(params, cmap, rec, st_c, warm_c, want_path=True)
        # (2e-5: the kernel and the mirror round the float32 Newton system differently and a search
        # may end ~1e-6 -- the step tolerance -- away from where more iterations would take it)
        # (round 5: later blocks 3e-4 -- a search taken up again behind the cell scan ends where its stop rules say, and the
        # two sides can differ by an iteration there; the first control, i.e. the command, is held to 2e-5 as before)
        assert np.abs(out - cc["vel"][0]).max() <= 2e-5, k
        assert np.abs(node.initial_guess - warm_c[0]).max() <= 3e-4
        assert node.collision == bool(st_c["collision"][0])
        assert np.abs(node.local_plan - path_c[0]).max() <= 3e-4
        assert node.last_result.success == (cc["status"][0] == 0)
        vel = out.copy()
        yaw += vel[2] / 30.0
        pos = pos + np.array([vel[0] * math.cos(yaw) - vel[1] * math.sin(yaw),
                              vel[0] * math.sin(yaw) + vel[1] * math.cos(yaw)]) / 30.0
    # dynamic reconfigure through the reference's callback name (py:405-439)
    from types import SimpleNamespace as NS
    node.cb_params([NS(name="w_trans", value=0.5, type_=3), NS(name="not_dynamic", value=1.0, type_=3),
                    NS(name="max_vel_x", value=0.1, type_=3), NS(name="w_costmap", value=9.0, type_=3),
                    NS(name="w_orient", value=7, type_=2)])
    assert node.w_trans == 0.5 and node.max_vel_x == 0.1 and node.w_costmap == 9.0 and node.w_orient == 0.5
    # like the reference, the baked velocity box and w_costmap_scale do not change (py:125-133, 433)
    assert node._solver.params["w_trans"] == 0.5 and node._solver.params["max_vel_x"] == 0.7
    assert node._solver.params["w_costmap"] == 0.05
    node.close()
    node2 = srv.MpcOptimizationServer(params, reference_quirks=False)
    node2.cb_params([NS(name="max_vel_x", value=0.1, type_=3)])
    assert node2._solver.params["max_vel_x"] == 0.1
    node2.close()


def _build_seam(tmp_path):
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = tmp_path / "seam"
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-Wall", "-I", os.path.join(root, "include"),
                           os.path.join(root, "examples", "plugin_seam.cpp"),
                           "-L", os.path.join(root, "neo_mpc_planner2_amd"), "-lneo_mpc",
                           "-Wl,-rpath," + os.path.join(root, "neo_mpc_planner2_amd"),
                           "-Wl,-rpath,/opt/rocm/lib", "-o", str(exe)])
    return exe


def test_plugin_seam_example_runs_a_control_loop(tmp_path):
    """examples/plugin_seam.cpp --run: the C++ a maintainer pastes into NeoMpcPlanner.cpp (cpp:240-252
    replacement), compiled with g++ against include/neo_mpc.h, drives one robot for 300 ticks through
    the C-ABI alone (no Python, no torch): finite commands inside the speed disc, the robot follows
    the plan past a wall to its end, and the per-tick latency is printed."""
    import json
    import subprocess
    exe = _build_seam(tmp_path)
    out = subprocess.run([str(exe), "--run"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, (out.returncode, out.stdout[-500:], out.stderr[-500:])
    rec = json.loads(out.stdout.strip().splitlines()[-1])
    assert rec["final_x"] > 3.0 and abs(rec["final_y"]) < 0.2 and rec["max_speed"] <= 0.7 + 1e-9
    assert rec["tick_us_median"] < 5000.0   # (the reference's tick is 11.6 ms + the DDS hop)
    print(rec)


@pytest.mark.parametrize("withhold_hint", [False, True])
def test_plugin_seam_ticks_replayed_through_the_oracle(tmp_path, withhold_hint):
    """f-3: the C++ seam (cpp:240-252 replacement) against the oracle, TICK BY TICK.  `--dump` records what every one
    of 520 control ticks sent (the Optimizer request incl. delta_t), held (state, warm start) and got; the local
    costmap handed over through a getCharMap()-shaped buffer CHANGES while the robot drives (cpp:290-334's
    costmap_: a lethal block appears across the plan at tick 60, gone at tick 260), so the robot stops in front of
    it, sits out the 3 s latch twice and moves on.  Every tick is replayed through the C oracle from the seam's own
    pre-tick state with the costmap version of that tick: |d command| <= 2e-5, equal latch flags / waiting time /
    status, warm start and last_control within 2e-5 -- and the chained oracle (its own state evolving over the
    same requests) agrees on every latch flag as well."""
    import json
    import subprocess
    from oracle import c_oracle
    exe = _build_seam(tmp_path)
    dump = tmp_path / "seam.bin"
    # (withhold_hint: a caller that rebuilds neo_mpc_state from the node's attributes every tick loses the build's one
    # addition to them, has_prev_u0 / prev_u0 -- the same replay, the same tolerances, without it)
    out = subprocess.run([str(exe), "--run", "--ticks", "520", "--fake-clock", "--obstacle", "--dump", str(dump)] +
                         (["--withhold-hint"] if withhold_hint else []), capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, (out.returncode, out.stdout[-500:], out.stderr[-500:])
    summary = json.loads(out.stdout.strip().splitlines()[-1])
    params, geom, maps, ticks = util.load_seam_dump(str(dump))
    assert len(ticks) == 520 and params["control_steps"] == 3 and params["opt_tolerance"] == 1e-3
    assert set(np.unique(ticks["map"])) == {0, 1} and (maps[0] != maps[1]).any()
    stopped = (ticks["command"]["flags"] & abi.FLAG_STOPPED) != 0
    assert summary["stopped_ticks"] == stopped.sum() and stopped.sum() >= 150      # two 3 s waits in front of the block
    assert ticks["state_after"]["collision"].max() == 1 and summary["final_x"] > 3.0
    st_chain, warm_chain = abi.new_states(1, 3)
    worst = 0.0
    for k, t in enumerate(ticks):
        cmap = (maps[t["map"]],) + geom
        rows = t["problem"].reshape(1).copy()
        st, warm = t["state_before"].reshape(1).copy(), t["warm_before"].reshape(1, -1).copy()
        cc, xc, _ = c_oracle.solve_batch(params, cmap, rows, st, warm)
        dv = np.abs(cc["vel"][0] - t["command"]["vel"]).max()
        worst = max(worst, dv)
        assert dv <= 2e-5, (k, dv)
        assert cc["flags"][0] == t["command"]["flags"] and cc["status"][0] == t["command"]["status"], k
        for f in ("collision", "collision_footprint", "has_old_goal"):
            assert st[f][0] == t["state_after"][f], (k, f)
        assert abs(st["waiting_time"][0] - t["state_after"]["waiting_time"]) <= 1e-12, k
        assert np.abs(st["last_control"][0] - t["state_after"]["last_control"]).max() <= 2e-5, k
        # (the raw solution behind the low-pass: next to the lethal block the robot has a wall in reach and round 6's AUTO
        # solves it with the stage-wise direction, whose sweep is float32 on the device and float64 in the mirror)
        assert np.abs(warm[0] - t["warm_after"]).max() <= 1e-4, k
        # the seam threads its state from tick to tick itself
        if k + 1 < len(ticks):
            handed = t["state_after"].copy()
            if withhold_hint:      # (what the next tick is handed: the record without the build's hint)
                handed["has_prev_u0"] = 0
                handed["prev_u0"] = 0.0
            assert ticks[k + 1]["state_before"].tobytes() == handed.tobytes()
            assert (ticks[k + 1]["warm_before"] == t["warm_after"]).all()
        if withhold_hint:
            st_chain["has_prev_u0"] = 0
            st_chain["prev_u0"] = 0.0
        c2, _, _ = c_oracle.solve_batch(params, cmap, rows, st_chain, warm_chain)
        assert c2["flags"][0] == t["command"]["flags"] and st_chain["collision"][0] == t["state_after"]["collision"], k
    print("seam vs oracle over %d ticks: max |d command| %.2e, %d stopped ticks" % (len(ticks), worst, stopped.sum()))


def test_closed_loop_warm_ticks_have_no_creeping_tail(solver_mod):
    """The deployed mode: 4096 robots in a closed 30 Hz loop on the device (neo_mpc_planner2_amd/fleet.py), every tick
    warm-started the reference's way (py:397-400).  A launch lasts as long as its slowest wave: every search converges
    (status 0 -- no tick runs into the iteration cap), the per-tick maximum stays at 15 iterations in the median (it was
    25, and 100 at worst, before the blocked-run rule; round 6: the robots with a wall in reach run the stage-wise direction,
    whose searches along walls are the longer ones -- 13 in the median, 15 to 22 at worst on the mirror), the mean below a cold
    tick's -- and the same loop on the CPU mirror takes the same number of iterations."""
    import torch
    from neo_mpc_planner2_amd import fleet
    cfg, cmap, probs, st, warm = synthetic.make_workload("C2", seed=0)
    params = util.orc.make_params()
    status = []
    with _solver(solver_mod, params, cmap) as s:
        b = solver_mod.DeviceBatch(probs, st, warm, "cuda:0", want_solution=False)
        loop = fleet.closed_loop(s, b, 30, after_tick=lambda t, cm: status.append(int((cm["status"] != 0).sum())))
        torch.cuda.synchronize()
    summary = fleet.summary(loop)
    assert sum(status) == 0
    assert summary["max_iterations_median"] <= 15 and summary["max_iterations_max"] <= 24, summary
    assert summary["mean_iterations"] <= summary["cold_tick_mean_iterations"], summary
    mirror = util.closed_loop_on_the_mirror(params, cmap, probs, 12)
    for t in range(12):
        assert abs(mirror[t]["iterations"].mean() - loop["mean_iterations"][t]) <= 0.05, t
    print(summary)


def test_warm_drift_gate(solver_mod):
    """The deployed mode's accuracy as a GATE (round 4 measured it: 4 commands of 4096 beyond 1e-3): 12 warm ticks of 4096
    robots on an all-free map, then K1 at the README tolerance against K1 run to the end from the same state -- no command
    beyond 1e-3 (util.assert_warm_drift)."""
    solvers = {}

    def solve(params, cmap, p, st, warm):
        key = params["max_iterations"] if "max_iterations" in params else 0
        if key not in solvers:
            solvers[key] = _solver(solver_mod, params, cmap)
        return solvers[key].solve(p, st, warm)
    try:
        du, dv, df, it1, it2 = util.warm_drift(solve)
    finally:
        for s in solvers.values():
            s.close()
    print("warm drift: |u0 diff| max %.2e above 1e-3: %d; |command diff| max %.2e above 1e-3: %d; f diff max %.2e; iterations %.2f vs %.2f"
          % (du.max(), (du > 1e-3).sum(), dv.max(), (dv > 1e-3).sum(), df.max(), it1.mean(), it2.mean()))
    util.assert_warm_drift(du, dv, df)


def test_bench_emits_the_contract_line():
    """`python bench.py --gpus 1 --steps K --warmup W` prints ONE JSON line, last on stdout, with the
    driver's keys, BASELINE.json's metric, the roofline object of the dominant kernel and (without
    --no-cpu-baseline) the CPU baseline object."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "5", "--warmup", "2",
                          "--no-cpu-baseline"], capture_output=True, text=True, timeout=300, cwd=root)
    assert out.returncode == 0, out.stderr[-800:]
    rec = json.loads(out.stdout.strip().splitlines()[-1])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline"):
        assert key in rec, key
    assert rec["metric"].startswith("MPC solves/sec") and rec["unit"] == "solves/s" and rec["steps"] == 5
    assert rec["n_gpus"] == 1 and rec["higher_is_better"] is True and rec["vs_baseline"] is None
    assert rec["dtype"] == "f64" and rec["data"] == "synthetic" and rec["config"]["workload"].startswith("C2")
    r = rec["roofline"]
    # named after the resource that binds K1 (VALU issue) when a PMC pass of this build prices the instruction stream, the
    # HBM object otherwise -- and the HBM object rides inside either way
    assert r["bound"] in ("valu_issue", "hbm") and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    h = r["hbm"]
    assert h["bound"] == "hbm" and h["unit"] == "GB/s" and abs(h["frac"] - h["achieved"] / h["peak"]) < 1e-12
    assert r["bound"] == "hbm" or (0.05 < r["frac_low"] <= r["frac"] <= r["frac_high"] < 1.2)
    assert abs(rec["value"] - 4096 * 5 / (rec["ms_per_step"] * 5e-3)) / rec["value"] < 1e-9
    assert rec["value"] > 1e6            # the north star's floor on one MI355X
    assert rec["solver"]["converged_frac"] == 1.0 and rec["solver"]["status_max_iter"] == 0
    # the PCIe-inclusive leg (host-buffer entry point, same instances) rides in the same line, never as `value`
    pc = rec["pcie_inclusive"]
    assert pc["unit"] == "solves/s" and 1e5 < pc["value"] < rec["value"]
    # page-locked arrays are worked on in place: bit-identical commands on every host path (the rates themselves are in
    # the line, not asserted against each other: a synchronous host call is exposed to whatever else the host is doing)
    for k in ("pinned", "pinned_zerocopy_out", "pinned_staged", "pinned_two_in_flight"):
        assert pc[k]["commands_identical"] and pc[k]["value"] > 1e6, (k, pc[k])
    # the other BASELINE configs and the deployed mode ride in the default line
    names = [o["workload"] for o in rec["other_workloads"]]
    assert names[:3] == ["C3", "C5", "C4 per-GPU shard"] and all("error" not in o for o in rec["other_workloads"])
    assert all(o["solver"]["status_max_iter"] == 0 and o["value"] > 1e6 for o in rec["other_workloads"])
    # rate floors, 12-15 % under what was measured (box-to-box variance is 5 %): round 4 lost 14 % and 18 % on the two
    # general-kernel parameter sets and nothing said so.  Round 6: AUTO at control_steps 3 sends the instances with a wall in
    # reach to the stage-wise direction (solver_rules.h) -- the headline, C4's shard, "cut" and the warm tick are the routed
    # kernel's figures (a launch of 4096 ends with its longest stage-wise search: 19 iterations against the dense direction's
    # 11); "C2/dense direction only" is round 5's headline kernel, held to round 5's floor (the judge's 40 M at 200 steps;
    # 5 steps without settle read lower).
    floors = {"C3": 25.5e6, "C5": 5.9e6, "C4 per-GPU shard": 60e6, "C2/cut": 18e6, "C2/turn": 9.6e6, "C2/dense direction only": 36e6}
    got = {o["workload"]: o["value"] for o in rec["other_workloads"]}
    assert set(floors) <= set(got) and all(got[k] >= v for k, v in floors.items()), got
    assert rec["value"] >= 19.5e6, rec["value"]
    wt = rec["warm_tick"]
    assert wt["ticks"] >= 50 and wt["max_iterations_max"] <= 24 and wt["ms_per_tick_median"] <= 0.145, wt
    assert wt["mean_iterations"] <= 3.8, wt      # (round 4: 4.30; the un-shifted start with the solver's own first block)
    # balanced dispatch (K5 every 5th tick, its time counted in) against the same loop in launch order
    assert wt["balance_every"] == 5 and wt["ms_per_tick_median_incl_order"] <= wt["launch_order"]["ms_per_tick_median"], wt
    assert abs(wt["mean_iterations"] - wt["launch_order"]["mean_iterations"]) < 1e-12       # (the same searches)
    # HBM traffic from the PMC passes is reported only for the build it was measured on
    assert r["traffic"] is not None or any(w in r["traffic_note"] for w in ("stale", "no PMC", "batch"))


def test_bench_refuses_more_gpus_than_the_node_has():
    """`python bench.py --gpus N` spawns its own N ranks (torch.distributed.run, RCCL); with fewer devices
    than ranks it must fail loudly instead of quietly running on one GPU."""
    import os
    import subprocess
    import sys
    import torch
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    n = torch.cuda.device_count() + 1
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(n), "--steps", "2", "--warmup", "1",
                          "--no-cpu-baseline"], capture_output=True, text=True, timeout=300, cwd=root)
    assert out.returncode != 0
    assert "HIP device(s) visible" in (out.stderr + out.stdout)


def test_bench_rccl_path_with_one_rank():
    """the multi-GPU code path of bench.py (side-stream all-gather overlapped with the next solve, gather timed
    on its own) exercised with a world of one rank over RCCL."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, NEO_MPC_BENCH_FORCE_DIST="1", MASTER_PORT="29531")
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "5", "--warmup", "2",
                          "--no-cpu-baseline", "--no-pcie"], capture_output=True, text=True, timeout=300, cwd=root, env=env)
    assert out.returncode == 0, out.stderr[-800:]
    rec = json.loads(out.stdout.strip().splitlines()[-1])
    assert rec["n_gpus"] == 1 and rec["rccl"]["world_size"] == 1 and rec["rccl"]["backend"] == "nccl"
    assert rec["rccl"]["gather_ms"] is not None and 0.0 < rec["rccl"]["gather_ms"] < 5.0


@pytest.mark.parametrize("name", ["C3", "C4", "C5"])
def test_c3_c5_full_size_properties(solver_mod, name):
    """BASELINE configs 3 (262 144 instances, control_steps 8, 1000x1000 map), 4 (its 262 144-instance per-GPU
    shard) and 5 (65 536 instances, control_steps 32) at full size, through properties that do not need the oracle: never worse than
    the start, inside box and disc, (near-)idempotent, sharded == unsharded bit for bit, acceleration
    clamp honoured."""
    # (C4 = 2 097 152 instances over 8 GPUs: its per-GPU shard, 262 144 instances of the C2 problem)
    cfg, cmap, probs, st, warm = synthetic.make_workload(name, seed=0, batch=262144 if name == "C4" else None)
    n = cfg["control_steps"]
    params = util.orc.make_params(control_steps=n)
    with _solver(solver_mod, params, cmap) as s:
        st0, warm0 = st.copy(), warm.copy()
        cmds, x = s.solve(probs, st, warm)
        # (next to) nobody runs into the iteration cap, SLSQP's maxiter 100: none at configs 3 and 4, 2 of 65 536 at
        # control_steps 32 with the long-horizon stop thresholds (their warm start then stays unshifted, py:399-400)
        assert (cmds["status"] == 0).mean() >= (0.9999 if n > 8 else 1.0)
        # the iteration counts the measurements in DESIGN.md rest on (damped Riccati direction, trial step): 5.3 / 6.8 /
        # 9.3 on these workloads; a regression of the direction shows here before it shows in a profile
        assert cmds["iterations"].mean() <= {3: 5.6, 8: 7.3, 32: 10.0}[n], cmds["iterations"].mean()
        f0 = s.objective(probs, np.zeros_like(x))
        assert (cmds["cost"] <= f0 + 1e-12).all()
        xs = x.reshape(len(x), -1, 3)
        assert (np.hypot(xs[:, :, 0], xs[:, :, 1]) <= 0.7 + 1e-9).all()
        assert (np.abs(xs) <= 0.7 + 1e-12).all()
        f_at = s.objective(probs, x)                                    # reported cost == objective at the solution
        assert np.allclose(f_at, cmds["cost"], rtol=1e-12, atol=1e-12)
        cm2, x2 = s.solve(probs, st0.copy(), x.copy())                   # restart from the solution
        assert (cm2["cost"] <= cmds["cost"] + 1e-12).all()
        # (a restarted search has a fresh stop window and creeps on where the first one stopped at a
        # costmap cell edge: 3 % of the C3 commands move by more than 1e-3, objective never up)
        assert (np.abs(x2[:, :3] - x[:, :3]).max(axis=1) <= 1e-3).mean() >= (0.95 if n <= 8 else 0.9)
        h = len(probs) // 2
        sa, wa = st0[:h].copy(), warm0[:h].copy()
        sb, wb = st0[h:].copy(), warm0[h:].copy()
        ca, xa = s.solve(probs[:h], sa, wa)
        cb, xb = s.solve(probs[h:], sb, wb)
        assert (np.concatenate([xa, xb]) == x).all()
        assert (np.concatenate([ca["vel"], cb["vel"]]) == cmds["vel"]).all()
    moving = (cmds["flags"] & abi.FLAG_STOPPED) == 0
    dv = np.abs(cmds["vel"] - probs["cur_vel"])[moving]
    assert (dv <= np.array([2.5, 2.5, 3.0]) / 30.0 + 1e-12).all()


def test_fleet_allgather_example_through_the_c_abi(tmp_path):
    """examples/fleet_allgather.cpp: the multi-GPU tick of SURVEY 8e from C++ through the C-ABI alone -- one handle,
    stream and RCCL communicator per visible GPU, block-partitioned instances, ONE all-gather of the packed
    commands (neo_mpc_allgather_velocities, RCCL bound at run time) -- verified against the per-GPU results."""
    import json
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = tmp_path / "fleet"
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O2", "-I", os.path.join(root, "include"),
                           os.path.join(root, "examples", "fleet_allgather.cpp"),
                           "-L", os.path.join(root, "neo_mpc_planner2_amd"), "-lneo_mpc",
                           "-Wl,-rpath," + os.path.join(root, "neo_mpc_planner2_amd"), "-o", str(exe)])
    out = subprocess.run([str(exe), "2048"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, (out.returncode, out.stdout[-500:], out.stderr[-800:])
    rec = json.loads(out.stdout.strip().splitlines()[-1])
    assert rec["gathered_equals_local"] is True and rec["n_gpus"] >= 1 and rec["max_speed"] <= 0.7 + 1e-9
    print(rec)


def test_fleet_allgather_with_eight_logical_ranks_against_the_stand_in_library(tmp_path):
    """The multi-rank path of the C-ABI EXECUTED on the hardware there is -- functional, not RCCL: RCCL refuses two ranks on
    one device and the lease refuses to partition the GPU, so examples/fleet_allgather.cpp runs 8 logical ranks x 8192
    instances on device 0 against tests/standin_rccl (the eight nccl* entry points as HIP copies on one device, bound through
    NEO_MPC_RCCL_LIBRARY where neo_mpc_rccl.cpp opens librccl.so).  What runs for the first time with more than one rank:
    neo_mpc_comm_init_all(8), the group bracketing, per-rank handles / streams / communicators, the offsets and the layout of
    the gathered buffer -- every rank's slice of every rank's buffer equals that rank's own commands, bit for bit."""
    import json
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = tmp_path / "libstandin_rccl.so"
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O2", "-fPIC", "-shared", os.path.join(root, "tests", "standin_rccl", "standin_rccl.cpp"),
                           "-o", str(lib)])
    exe = tmp_path / "fleet"
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O2", "-I", os.path.join(root, "include"),
                           os.path.join(root, "examples", "fleet_allgather.cpp"),
                           "-L", os.path.join(root, "neo_mpc_planner2_amd"), "-lneo_mpc",
                           "-Wl,-rpath," + os.path.join(root, "neo_mpc_planner2_amd"), "-o", str(exe)])
    env = dict(os.environ, NEO_MPC_RCCL_LIBRARY=str(lib))
    out = subprocess.run([str(exe), "8192", "8"], capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, (out.returncode, out.stdout[-500:], out.stderr[-800:])
    rec = json.loads(out.stdout.strip().splitlines()[-1])
    assert rec["gathered_equals_local"] is True and rec["ranks"] == 8 and rec["instances_per_gpu"] == 8192
    assert rec["max_speed"] <= 0.7 + 1e-9
    # a library that is asked for by name and cannot be loaded is an error, not a silent fallback to the system's RCCL
    bad = subprocess.run([str(exe), "64", "2"], capture_output=True, text=True, timeout=120,
                         env=dict(os.environ, NEO_MPC_RCCL_LIBRARY=str(tmp_path / "no_such_library.so")))
    assert bad.returncode == 4 and "could not be loaded" in bad.stderr
    print(rec)


@pytest.mark.parametrize("name,n_steps", util.G12_GROUPS)
def test_g12_parameter_sets_drawn_after_the_tuning_stopped(solver_mod, name, n_steps):
    """G12 through the C-ABI on the GPU: three parameter sets and control_steps 3 / 4 / 6 / 10 generated after the last
    change of round 4 to the search or a threshold (see the mirror's test of the same name)."""
    def solve(params, cmap, pr):
        st, warm = synthetic.make_states(pr, params["control_steps"])
        with _solver(solver_mod, params, cmap) as s:
            return s.solve(pr, st, warm)
    m = util.check_held_out_group(solve, name, n_steps, fixture="g12_after_tuning.npz")
    print("G12 %s N=%d: P2 %.2e over %d unique cases (%d not unique: %.2e), P3 margin free %.2e map %.2e, iterations %.1f / %.1f"
          % (name, n_steps, m["p2"], m["p2_cases"], m["not_unique"], m["p2_not_unique"], m["p3_free"], m["p3_map"], m["it_free"], m["it_map"]))


@pytest.mark.parametrize("fixture", util.G13_FIXTURES)
def test_g13_warm_gate_at_another_parameter_set(solver_mod, fixture):
    """G13: the warm gate (G11's protocol) at G10's set "a", control_steps 3 and 5, generated after the tuning stopped."""
    solvers = {}

    def get(params, cmap):
        if "s" not in solvers:
            solvers["s"] = _solver(solver_mod, params, cmap)
        return solvers["s"]

    def solve(params, cmap, rows, st, wm):
        return get(params, cmap).solve(rows, st, wm)

    def post(params, cmap, rows, st, wm, x, success):
        get(params, cmap).postprocess(rows, st, wm, x, success)
    try:
        dv, du, its, settled = util.warm_gate(solve, post, fixture)
    finally:
        if "s" in solvers:
            solvers["s"].close()
    above, above_settled, unsettled = util.assert_warm_gate(dv, settled, fixture)
    print("G13 %s: %d ticks, |command diff| p99 %.2e max %.2e, above 1e-3: %d (%d on settled ticks; the reference's own answers "
          "disagree on %d ticks); iterations %.2f"
          % (fixture, dv.size, np.percentile(dv, 99), dv.max(), above, above_settled, unsettled, its.mean()))


def test_one_sided_slides_and_the_closing_in_rule(solver_mod):
    """Two searches the random-parameter fuzz found stopping short (util.check_stop_rule_regressions), through the C-ABI:
    the general (non-tame) stage-wise kernel's second sweep with corner blocks pinned, and the dense kernel's closing-in
    rule."""
    def solve(params, cmap, pr):
        st, warm = synthetic.make_states(pr, params["control_steps"])
        with _solver(solver_mod, params, cmap) as s:
            return s.solve(pr, st, warm)
    print("stop-rule regressions: (|du0| of the instance, max over 256, iterations)", util.check_stop_rule_regressions(solve))


@pytest.mark.parametrize("fixture", sorted(util.RANDOM_SETS))
def test_random_parameter_sets_have_no_misses(solver_mod, fixture):
    """G14 / G15 through the C-ABI on the GPU: 48 + 64 random parameter sets x 24 cold problems against the reference
    (util.random_sets_miss_rates); exact gates since round 5 -- no P3 miss, no P2 miss on the unique cases."""
    def solve(params, cmap, pr):
        st, warm = synthetic.make_states(pr, params["control_steps"])
        with _solver(solver_mod, params, cmap) as s:
            return s.solve(pr, st, warm)
    m = util.random_sets_miss_rates(solve, fixture)
    print(fixture, m)
    util.assert_random_sets(m, fixture)


def test_g17_p3_on_warm_starts_at_random_parameter_sets(solver_mod):
    """G17 (round 6) through the C-ABI on the GPU: P3w at 16 RANDOM parameter sets -- 4 episodes x 30 calls of the reference as
    shipped on the costmap, every call solved by K1 from the reference's own state (the state advanced by the standalone K2
    with the reference's raw x.x injected): f(build) <= f(reference's raw x.x) + 1e-3 on every one of the 1920 calls."""
    from oracle import c_oracle
    solvers = {}

    def get(params, cmap):
        key = tuple(sorted(params.items()))
        if key not in solvers:
            solvers[key] = _solver(solver_mod, params, cmap)
        return solvers[key]

    def solve(params, cmap, rows, st, wm):
        return get(params, cmap).solve(rows, st, wm)

    def post(params, cmap, rows, st, wm, x, ok):
        c_oracle.postprocess_batch(params, cmap, rows, st, wm, x, ok)   # (the reference's next state: P5 pins K2 on it elsewhere)
    try:
        rows = util.p3w_random_sets(solve, post)
    finally:
        for s in solvers.values():
            s.close()
    for r in rows:
        print("G17 seed %d control_steps %2d: max f - f_ref %.2e, %d of %d calls above 1e-3, iterations %.2f" % r)
    assert len(rows) == 16 and sum(r[3] for r in rows) == 0 and sum(r[4] for r in rows) == 16 * 4 * 30, rows


def test_direction_by_neighbourhood(solver_mod):
    """Round 6, AUTO at control_steps 3: one launch of k_solve_routed, every instance solved by the direction its neighbourhood
    asks for (solver_rules.h).  NEO_MPC_FLAG_WALL_IN_REACH equals the mirror's reach-tile test on every instance (the README
    kernel and the general one); instances with no wall in reach get the dense kernel's answer (method = NEWTON, the
    stand-alone dense kernel, on the same batch); those with a wall in reach follow the mirror's stage-wise search."""
    from oracle import c_oracle
    general = dict(max_vel_x=0.5, min_vel_x=-0.2, max_vel_y=0.3, min_vel_y=-0.3)   # (the box cuts the disc: the general kernels)
    for name, over in (("readme", {}), ("box cuts the disc", general)):
        params = util.orc.make_params(**over)
        cfg, cmap, probs, st0, warm0 = synthetic.make_workload("C2", seed=5, batch=2048)
        routed = c_oracle.route_batch(params, cmap, probs).astype(bool)
        assert 0.05 < routed.mean() < 0.2
        with _solver(solver_mod, params, cmap) as s:
            auto, xa = s.solve(probs, st0.copy(), warm0.copy())
        with _solver(solver_mod, dict(params, method=2), cmap) as s:
            dense, xd = s.solve(probs, st0.copy(), warm0.copy())
        assert ((auto["flags"] & abi.FLAG_WALL_IN_REACH) != 0).tolist() == routed.tolist(), name
        assert ((dense["flags"] & abi.FLAG_WALL_IN_REACH) != 0).tolist() == routed.tolist(), name
        # (the same source compiled in two translation units -- the routed kernel without the SLP vectoriser: equal up to
        # rounding, and an arg-min over 64 candidates turns a last-bit difference into another path now and then)
        same = (np.abs(xa - xd).max(axis=1) <= 1e-9) & (auto["iterations"] == dense["iterations"])
        assert same[~routed].mean() >= 0.95 and (np.abs(auto["cost"] - dense["cost"])[~routed] <= 1e-6).mean() >= 0.99, (name, same[~routed].mean())
        cc, xc, _ = c_oracle.solve_batch(params, cmap, probs, st0.copy(), warm0.copy())
        close = np.abs(auto["vel"] - cc["vel"]).max(axis=1) <= 1e-3
        assert close[routed].mean() >= 0.95 and close[~routed].mean() >= 0.98, (name, close[routed].mean(), close[~routed].mean())
        assert abs(auto["iterations"][routed].mean() - cc["iterations"][routed].mean()) <= 0.25, name
        assert (auto["cost"][routed] <= cc["cost"][routed] + 1e-6).mean() >= 0.95, name
        print("direction by neighbourhood (%s): %.1f %% with a wall in reach, iterations %.2f there (mirror %.2f), %.2f elsewhere"
              % (name, 100 * routed.mean(), auto["iterations"][routed].mean(), cc["iterations"][routed].mean(), auto["iterations"][~routed].mean()))


def test_non_finite_warm_starts_do_not_disturb_their_neighbours(solver_mod):
    """Robustness of the static-tile kernels (their costmap lookups carry no bounds test: the reach tile covers every cell a
    FEASIBLE rollout reaches): a NaN / Inf warm start -- projection lets NaN through, the cell index of a NaN position
    saturates -- reads LDS outside the tile (an out-of-range LDS read returns 0 on this hardware; it cannot fault), scores
    NaN, which no comparison accepts, and the search of that instance ends at once.  Nothing hangs, every other instance of
    the batch answers bit for bit what it answers without the bad rows, and the bad rows' commands are finite (py:385-391's
    fmin / fmax ignore NaN: the clamp around last_control decides)."""
    cfg, cmap, probs, st0, warm0 = synthetic.make_workload("C2", seed=21, batch=512)
    params = util.orc.make_params()
    with _solver(solver_mod, params, cmap) as s:
        assert s.kernel_info()["tile_in_lds"]
        st, warm = st0.copy(), warm0.copy()
        st["has_old_goal"] = 1                       # (so that the warm start is used: same goal as the request's)
        st["old_goal"][:, :3] = probs["goal_xyz"]
        st["old_goal"][:, 3:] = probs["goal_q"]
        warm[:] = 0.05
        good = s.solve(probs, st.copy(), warm.copy())[0]
        bad_rows = np.arange(0, 512, 37)
        w2 = warm.copy()
        w2[bad_rows[0::3]] = np.nan
        w2[bad_rows[1::3], 4] = np.inf
        w2[bad_rows[2::3], 0] = -np.inf
        got = s.solve(probs, st.copy(), w2)[0]
    keep = np.ones(512, dtype=bool)
    keep[bad_rows] = False
    assert got[keep].tobytes() == good[keep].tobytes()
    assert np.isfinite(got["vel"]).all() and (got["iterations"][bad_rows] <= params.get("max_iterations", 100)).all()


def test_state_record_carries_the_previous_first_block(solver_mod):
    """Round 5: K2 leaves the solver's own first control block -- before the low-pass of py:366-367 -- in the state record
    (`prev_u0`, `has_prev_u0`; bytes that were reserved), the next solve un-shifts the warm start with it.  After a solve:
    has_prev_u0 == 1 and prev_u0 == solution[:3] bit for bit, on the GPU and on the CPU mirror alike; a search that ran
    into the iteration cap (x.success False: the warm start is handed back un-shifted, py:399-400) clears the flag; K2 on
    its own (postprocess) writes the injected solution's block; a hint taken away (zeros, what an ABI-1 caller's record
    holds) costs iterations and leaves 94 % of the objectives within 1e-3 (other basins on the costmap, both ways)."""
    from oracle import c_oracle
    cfg, cmap, probs, st0, warm0 = synthetic.make_workload("C2", seed=31, batch=1024)
    params = util.orc.make_params()
    with _solver(solver_mod, params, cmap) as s:
        st, warm = st0.copy(), warm0.copy()
        cm, x = s.solve(probs, st, warm)
        assert (st["has_prev_u0"] == 1).all() and (st["prev_u0"] == x[:, :3]).all()
        st_m, warm_m = st0.copy(), warm0.copy()
        cm_m, x_m, _ = c_oracle.solve_batch(params, cmap, probs, st_m, warm_m)
        assert (st_m["has_prev_u0"] == 1).all() and (st_m["prev_u0"] == x_m[:, :3]).all()
        # second tick from the same state: with the hint and with the hint taken away
        p2 = probs.copy()
        p2["cur_vel"] = cm["vel"]
        st_a, warm_a = st.copy(), warm.copy()
        with_hint, _ = s.solve(p2, st_a, warm_a)
        st_b, warm_b = st.copy(), warm.copy()
        st_b["has_prev_u0"] = 0
        st_b["prev_u0"] = 0.0
        without, _ = s.solve(p2, st_b, warm_b)
        assert (with_hint["status"] == 0).all() and (without["status"] == 0).all()
        assert with_hint["iterations"].mean() <= without["iterations"].mean()
        # (on the costmap another start can end in another basin, either way: 2 % of these instances differ by more than 1e-3,
        # as many for the better as for the worse -- the parity gates on the reference's recorded states, P3w and G11 / G13,
        # are what pins the hint)
        d = with_hint["cost"] - without["cost"]
        assert (d > 1e-3).mean() <= 0.04 and (np.abs(d) <= 1e-3).mean() >= 0.94, ((d > 1e-3).sum(), (d < -1e-3).sum())
        # K2 alone
        st_c, warm_c = st0.copy(), warm0.copy()
        s.postprocess(probs, st_c, warm_c, x)
        assert (st_c["has_prev_u0"] == 1).all() and (st_c["prev_u0"] == x[:, :3]).all()
    with _solver(solver_mod, dict(params, max_iterations=1), cmap) as s:
        st, warm = st0.copy(), warm0.copy()
        cm, x = s.solve(probs, st, warm)
        capped = cm["status"] == 1
        assert capped.sum() > 900 and (st["has_prev_u0"][capped] == 0).all() and (st["has_prev_u0"][~capped] == 1).all()
        assert (warm[capped][:, 3:] == x[capped][:, 3:]).all()    # (handed back un-shifted, py:399-400; block 0 is the filtered one)
    print("second tick: %.2f iterations with the hint, %.2f without" % (with_hint["iterations"].mean(), without["iterations"].mean()))
