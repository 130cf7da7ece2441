"""GPU edge cases of the hot path through the C-ABI: empty and tiny batches, control_steps 1 and
the maximum 64, a reach too large for the LDS tile (global-lookup path), robots outside / at the
edge of the map, a 1x1 map, box-cuts-disc bounds, footprints with the maximum number of points,
dynamic reconfigure, and multi-tick device-resident state."""
import numpy as np
import pytest

from neo_mpc_planner2_amd import abi, synthetic
from tests import util

pytestmark = pytest.mark.gpu


def _mirror(params, cmap, probs, st, warm, **kw):
    from oracle import c_oracle
    return c_oracle.solve_batch(params, cmap, probs, st, warm, **kw)


def _close(cg, cc, frac=0.97, tol=1e-3):
    dv = np.abs(cg["vel"] - cc["vel"]).max(axis=1)
    assert (dv <= tol).mean() >= frac, (dv <= tol).mean()


def test_empty_and_single_instance_batches():
    from neo_mpc_planner2_amd.solver import BatchSolver
    params = util.orc.make_params()
    cmap = synthetic.make_costmap(200, seed=1)
    with BatchSolver(params) as s:
        s.set_costmap(*cmap)
        st, warm = abi.new_states(0, 3)
        cmds, x = s.solve(np.zeros(0, dtype=abi.PROBLEM_DTYPE), st, warm)
        assert len(cmds) == 0 and x.shape == (0, 9)
        probs = synthetic.make_problems(1, 200, seed=2)
        st, warm = abi.new_states(1, 3)             # very first call: reset path (py:358-361)
        st["last_control"] = 0.4
        cmds, x = s.solve(probs, st, warm)
        assert cmds["flags"][0] & abi.FLAG_RESET
        st_c, warm_c = abi.new_states(1, 3)
        st_c["last_control"] = 0.4
        cc, xc, _ = _mirror(params, cmap, probs, st_c, warm_c)
        assert np.abs(cmds["vel"] - cc["vel"]).max() <= 1e-6
        assert st["has_old_goal"][0] == 1 and np.allclose(st["old_goal"][0][:3], probs["goal_xyz"][0])


@pytest.mark.parametrize("n_steps", [1, 2, 5, 64])
def test_unusual_control_steps(n_steps):
    """generic (runtime control_steps) kernel path, including the maximum 64 (192 variables)."""
    from neo_mpc_planner2_amd.solver import BatchSolver
    params = util.orc.make_params(control_steps=n_steps, max_iterations=100 if n_steps < 64 else 400)
    cmap = synthetic.make_costmap(200, seed=3)
    count = 128 if n_steps < 64 else 32
    probs = synthetic.make_problems(count, 200, seed=4 + n_steps)
    st, warm = synthetic.make_states(probs, n_steps)
    st_c, warm_c = st.copy(), warm.copy()
    with BatchSolver(params) as s:
        s.set_costmap(*cmap)
        cg, xg = s.solve(probs, st, warm)
        f0 = s.objective(probs, np.zeros_like(xg))
    assert (cg["cost"] <= f0 + 1e-12).all()
    cc, xc, _ = _mirror(params, cmap, probs, st_c, warm_c)
    _close(cg, cc, frac=0.9 if n_steps == 64 else 0.97)
    assert (cg["cost"] <= cc["cost"] + 1e-5).mean() >= 0.9


def test_reach_too_large_for_lds_tile_uses_global_lookups():
    from neo_mpc_planner2_amd.solver import BatchSolver
    # 2.5 m/s for 2 s on a 0.05 m grid = 100 cells of reach: no LDS tile
    params = util.orc.make_params(max_vel_x=2.5, min_vel_x=-2.5, max_vel_y=2.5, min_vel_y=-2.5, max_vel_trans=2.5,
                                  prediction_horizon=2.0)
    cmap = synthetic.make_costmap(500, seed=5)
    probs = synthetic.make_problems(256, 500, seed=6)
    st, warm = synthetic.make_states(probs, 3)
    st_c, warm_c = st.copy(), warm.copy()
    with BatchSolver(params) as s:
        s.set_costmap(*cmap)
        assert s.kernel_info()["tile_in_lds"] is False
        cg, xg = s.solve(probs, st, warm)
    cc, xc, _ = _mirror(params, cmap, probs, st_c, warm_c)
    _close(cg, cc, frac=0.95)
    # and the README configuration does stage the 27-row tile
    with BatchSolver(util.orc.make_params()) as s:
        s.set_costmap(*cmap)
        info = s.kernel_info()
        assert info["tile_in_lds"] and info["reach_cells"] == 13 and info["lds_bytes"] < 8192


def test_robots_outside_and_at_the_edge_of_the_map_and_tiny_map():
    from neo_mpc_planner2_amd.solver import BatchSolver
    params = util.orc.make_params()
    for cmap in (synthetic.make_costmap(200, seed=7), (np.zeros((1, 1), np.uint8), 0.05, 0.0, 0.0),
                 (np.full((3, 130), 17, np.uint8), 0.05, -3.25, -0.075)):
        half_x = cmap[0].shape[1] * cmap[1] / 2
        probs = synthetic.make_problems(64, 200, seed=8)
        probs["cur_xy"][:16] = np.random.default_rng(1).uniform(-1, 1, (16, 2)) * 0.2 + (cmap[2], cmap[3])
        probs["cur_xy"][16:32] += 50.0                      # far outside: every lookup is lethal
        probs["cur_xy"][32:48, 0] = cmap[2] + 2 * half_x - 0.01   # hugging the far edge
        st, warm = synthetic.make_states(probs, 3)
        st_c, warm_c = st.copy(), warm.copy()
        with BatchSolver(params) as s:
            s.set_costmap(*cmap)
            cg, xg = s.solve(probs, st, warm)
        cc, xc, _ = _mirror(params, cmap, probs, st_c, warm_c)
        assert (st["collision"] == st_c["collision"]).mean() >= 0.97
        _close(cg, cc, frac=0.95)
        assert (cg["vel"][16:32] == 0.0).all() and (st["collision"][16:32] == 1).all()


def test_box_cutting_the_disc():
    from neo_mpc_planner2_amd.solver import BatchSolver
    params = util.orc.make_params(max_vel_trans=0.7, max_vel_x=0.4, min_vel_x=-0.2, max_vel_y=0.65, min_vel_y=-0.65)
    probs = synthetic.make_problems(256, 200, seed=91)
    zero = (np.zeros((200, 200), np.uint8), 0.05, -5.0, -5.0)
    st, warm = synthetic.make_states(probs, 3)
    st_c, warm_c = st.copy(), warm.copy()
    with BatchSolver(params) as s:
        s.set_costmap(*zero)
        cg, xg = s.solve(probs, st, warm)
    xs = xg.reshape(len(xg), -1, 3)
    assert (xs[:, :, 0] <= 0.4 + 1e-12).all() and (xs[:, :, 0] >= -0.2 - 1e-12).all()
    assert (np.hypot(xs[:, :, 0], xs[:, :, 1]) <= 0.7 + 1e-9).all()
    cc, xc, _ = _mirror(params, zero, probs, st_c, warm_c)
    assert (np.abs(xg - xc).max(axis=1) <= 1e-4).mean() >= 0.97
    assert (cg["cost"] <= cc["cost"] + 1e-7).all()


def test_sixteen_point_footprints():
    from neo_mpc_planner2_amd.solver import BatchSolver
    from oracle import c_oracle
    cmap = synthetic.make_costmap(200, seed=9)
    probs = synthetic.make_problems(300, 200, seed=10)
    ang = np.linspace(0, 2 * np.pi, 16, endpoint=False)
    base = tuple((0.4 * np.cos(a), 0.3 * np.sin(a)) for a in ang)
    fps = np.array([synthetic.footprint_world(r, base) for r in probs])
    want = c_oracle.footprint_cost_batch(cmap, fps)
    st, warm = synthetic.make_states(probs, 3)
    with BatchSolver(util.orc.make_params(w_footprint=2000)) as s:
        s.set_costmap(*cmap)
        cg, xg = s.solve(probs, st, warm, footprints=fps)
        with pytest.raises(Exception):
            s.solve(probs, st, warm, footprints=np.zeros((300, 17, 2)))
    assert (st["collision_footprint"] == (want == 1.0)).all() and (want == 1.0).any()


def test_dynamic_reconfigure_takes_effect():
    from neo_mpc_planner2_amd.solver import BatchSolver
    cmap = synthetic.make_costmap(200, seed=11)
    probs = synthetic.make_problems(64, 200, seed=12)
    with BatchSolver(util.orc.make_params()) as s:
        s.set_costmap(*cmap)
        st, warm = synthetic.make_states(probs, 3)
        a, _ = s.solve(probs, st, warm)
        s.set_params(w_trans=0.1, max_vel_theta=0.2, min_vel_theta=-0.2, w_costmap=5.0)
        st, warm = synthetic.make_states(probs, 3)
        b, xb = s.solve(probs, st, warm)
        st_c, warm_c = synthetic.make_states(probs, 3)
        cc, _, _ = _mirror(s.params, cmap, probs, st_c, warm_c)
        with pytest.raises(ValueError):
            s.set_params(control_steps=4)
    assert (np.abs(xb.reshape(64, 3, 3)[:, :, 2]) <= 0.2 + 1e-12).all()
    assert np.abs(a["vel"] - b["vel"]).max() > 1e-3
    _close(b, cc)


def test_multi_tick_state_stays_resident_on_the_device():
    """20 control ticks for 512 robots with the records living in HBM (DeviceBatch): warm start,
    last_control, latch and goal bookkeeping carried by the kernel across launches == the CPU
    mirror fed tick by tick."""
    import torch
    from neo_mpc_planner2_amd.solver import BatchSolver, DeviceBatch
    params = util.orc.make_params()
    cmap = synthetic.make_costmap(500, seed=13)
    count = 512
    probs = synthetic.make_problems(count, 500, seed=14)
    st, warm = abi.new_states(count, 3)
    st_c, warm_c = st.copy(), warm.copy()
    pos = probs["cur_xy"].copy()
    yaw = np.random.default_rng(3).uniform(-3, 3, count)
    vel = np.zeros((count, 3))
    dt = 1.0 / 30.0
    agree = []
    with BatchSolver(params) as s:
        s.set_costmap(*cmap)
        db = DeviceBatch(probs, st, warm, "cuda:0")
        for k in range(20):
            if k == 10:
                g = synthetic.make_problems(count, 500, seed=99)
                probs["goal_xyz"][::2], probs["goal_q"][::2] = g["goal_xyz"][::2], g["goal_q"][::2]
            probs["cur_xy"], probs["cur_q"], probs["cur_vel"] = pos, synthetic.yaw_quat(yaw), vel
            probs["delta_t"] = dt if k else 1e9
            db.problems.copy_(torch.from_numpy(probs.view(np.uint8).reshape(count, -1)))
            s.solve_device(db.problems, db.states, db.warm, db.commands)
            torch.cuda.synchronize()
            cg = db.commands_host()
            cc, _, _ = _mirror(params, cmap, probs, st_c, warm_c)
            agree.append((np.abs(cg["vel"] - cc["vel"]).max(axis=1) <= 1e-3).mean())
            vel = cc["vel"].copy()           # both chains are driven by the same (mirror) commands
            yaw = yaw + vel[:, 2] * dt
            pos = pos + dt * np.stack([vel[:, 0] * np.cos(yaw) - vel[:, 1] * np.sin(yaw),
                                       vel[:, 0] * np.sin(yaw) + vel[:, 1] * np.cos(yaw)], 1)
        sg = db.states_host()
    assert min(agree) >= 0.95, agree
    assert (sg["collision"] == st_c["collision"]).mean() >= 0.97
    assert (sg["has_old_goal"] == 1).all()


@pytest.mark.parametrize("method,mem", [(0, 2), (0, 8), (1, 2), (1, 6), (1, 4)])
def test_lbfgs_memory_and_method_combinations_at_three_steps(method, mem):
    """control_steps 3 picks a compile-time LDS layout (4 pair slots) for the Newton and the
    memory-4 L-BFGS specialisations and the runtime layout otherwise: every combination must agree
    with the mirror."""
    from neo_mpc_planner2_amd.solver import BatchSolver
    params = util.orc.make_params(method=method, lbfgs_memory=mem)
    cmap = synthetic.make_costmap(500, seed=31)
    probs = synthetic.make_problems(512, 500, seed=32)
    st, warm = synthetic.make_states(probs, 3)
    st_c, warm_c = st.copy(), warm.copy()
    with BatchSolver(params) as s:
        s.set_costmap(*cmap)
        cg, xg = s.solve(probs, st, warm)
    cc, xc, _ = _mirror(params, cmap, probs, st_c, warm_c)
    _close(cg, cc, frac=0.97)
    assert (cg["cost"] <= cc["cost"] + 1e-6).mean() >= 0.97
    assert (cg["status"] == 0).mean() >= 0.99


def test_newton_is_refused_for_other_control_steps():
    from neo_mpc_planner2_amd import _lib
    from neo_mpc_planner2_amd.solver import BatchSolver
    with pytest.raises(_lib.NeoMpcError) as e:
        BatchSolver(util.orc.make_params(control_steps=9, method=2))
    assert e.value.code == -1 or "control_steps <= 8" in str(e.value)


def test_window_tolerance_trims_the_creeping_tail_only():
    """`window_tolerance` (three iterations gaining less than 3e-3 * opt_tolerance together end the
    search) trims instances that creep along a costmap cell edge: never more iterations, the
    objective within 1e-4 (<< opt_tolerance, SLSQP's own single-iteration test) of the run without
    it, the command unchanged on nearly every instance; < 0 switches it off, and both agree with
    the mirror run with the same setting."""
    from neo_mpc_planner2_amd.solver import BatchSolver
    cmap = synthetic.make_costmap(500, seed=41)
    probs = synthetic.make_problems(2048, 500, seed=42)
    out = {}
    for wt in (-1.0, 0.0):
        params = util.orc.make_params(window_tolerance=wt, method=2)   # (the dense direction's rule: every instance dense)
        st, warm = synthetic.make_states(probs, 3)
        st_c, warm_c = st.copy(), warm.copy()
        with BatchSolver(params) as s:
            s.set_costmap(*cmap)
            cg, xg = s.solve(probs, st, warm)
        cc, xc, _ = _mirror(params, cmap, probs, st_c, warm_c)
        _close(cg, cc, frac=0.97)
        assert abs(cg["iterations"].mean() - cc["iterations"].mean()) < 0.15
        out[wt] = cg
    off, on = out[-1.0], out[0.0]
    assert (on["iterations"] <= off["iterations"]).mean() >= 0.99
    assert on["iterations"].max() < off["iterations"].max()
    assert (on["cost"] <= off["cost"] + 1e-4).all()
    assert (np.abs(on["vel"] - off["vel"]).max(axis=1) <= 1e-3).mean() >= 0.99


@pytest.mark.parametrize("n_steps", [3, 8])
def test_disc_in_box_specialisation_equals_the_general_kernel(n_steps, monkeypatch):
    """README-like parameters (the max_vel_trans disc inside the vx/vy box, heading within pi/4 over
    the horizon) run kernels compiled without the box/disc corner cases and without the sin/cos range
    reduction; NEO_MPC_NO_TAME_SPECIALISATION selects the general ones: same results up to rounding."""
    from neo_mpc_planner2_amd.solver import BatchSolver
    params = util.orc.make_params(control_steps=n_steps)
    cmap = synthetic.make_costmap(500, seed=51)
    probs = synthetic.make_problems(512, 500, seed=52)
    res = []
    for general in (False, True):
        if general:
            monkeypatch.setenv("NEO_MPC_NO_TAME_SPECIALISATION", "1")
        st, warm = synthetic.make_states(probs, n_steps)
        with BatchSolver(params) as s:
            s.set_costmap(*cmap)
            res.append(s.solve(probs, st, warm))
    (c0, x0), (c1, x1) = res
    assert (c0["iterations"] == c1["iterations"]).mean() >= 0.97
    assert (np.abs(x0 - x1).max(axis=1) <= 1e-6).mean() >= 0.97   # (sin/cos differ in the last place)
    assert (np.abs(c0["vel"] - c1["vel"]).max(axis=1) <= 1e-6).mean() >= 0.99


def test_fast_turning_robot_uses_the_range_reduced_trigonometry():
    """max_vel_theta * prediction_horizon > pi/4: the heading can leave the range of the reduction-free
    sin/cos kernels, so the general kernels (Cody-Waite reduction) must be chosen -- results agree
    with the mirror (libm sin/cos) as for any other parameter set."""
    from neo_mpc_planner2_amd.solver import BatchSolver
    params = util.orc.make_params(max_vel_theta=3.0, min_vel_theta=-3.0, w_orient=2.0)
    cmap = synthetic.make_costmap(500, seed=61)
    probs = synthetic.make_problems(512, 500, seed=62)
    probs["carrot_q"] = synthetic.yaw_quat(np.random.default_rng(63).uniform(-3.0, 3.0, len(probs)))
    st, warm = synthetic.make_states(probs, 3)
    st_c, warm_c = st.copy(), warm.copy()
    with BatchSolver(params) as s:
        s.set_costmap(*cmap)
        cg, xg = s.solve(probs, st, warm)
    cc, xc, _ = _mirror(params, cmap, probs, st_c, warm_c)
    assert np.abs(xg[:, 2::3]).max() > 0.9          # the solutions do turn hard
    _close(cg, cc, frac=0.97)
    assert (cg["cost"] <= cc["cost"] + 1e-6).mean() >= 0.97


def test_static_lds_variant_equals_the_dynamic_one(monkeypatch):
    """The headline kernel keeps its LDS in a static array when the reach tile is small enough (every
    LDS address an instruction immediate); NEO_MPC_DYNAMIC_LDS selects the dynamic-LDS build of the
    same code: the same iterates up to the compiler's choice of fused multiply-adds."""
    from neo_mpc_planner2_amd.solver import BatchSolver
    params = util.orc.make_params()
    cmap = synthetic.make_costmap(500, seed=71)
    probs = synthetic.make_problems(512, 500, seed=72)
    res = []
    for dynamic in (False, True):
        if dynamic:
            monkeypatch.setenv("NEO_MPC_DYNAMIC_LDS", "1")
        st, warm = synthetic.make_states(probs, 3)
        with BatchSolver(params) as s:
            s.set_costmap(*cmap)
            res.append(s.solve(probs, st, warm))
    (c0, x0), (c1, x1) = res
    assert (c0["iterations"] == c1["iterations"]).mean() >= 0.99
    assert (np.abs(x0 - x1).max(axis=1) <= 1e-6).mean() >= 0.99
    assert (np.abs(c0["vel"] - c1["vel"]).max(axis=1) <= 1e-6).mean() >= 0.99
    assert np.allclose(c0["cost"], c1["cost"], rtol=0, atol=1e-6)


def test_costmap_pool_each_instance_reads_its_own_map():
    """Fleet variant (`neo_mpc_set_costmap_pool`): eight rolling windows of one geometry with their own
    contents and origins, instances spread over them by `map_index` -- every instance must come out as
    if it had been solved alone against its own map (mirror run per map), through the host and the
    device entry points; an out-of-range index is clamped."""
    import torch
    from neo_mpc_planner2_amd.solver import BatchSolver, DeviceBatch
    params = util.orc.make_params()
    m, size, count = 8, 160, 768
    rng = np.random.default_rng(81)
    maps = [synthetic.make_costmap(size, seed=90 + k) for k in range(m)]
    cells = np.stack([mp[0] for mp in maps])
    res = maps[0][1]
    offsets = rng.uniform(-50.0, 50.0, size=(m, 2))
    origins = np.array([[mp[2] + offsets[k, 0], mp[3] + offsets[k, 1]] for k, mp in enumerate(maps)])
    probs = synthetic.make_problems(count, size, seed=82)
    idx = rng.integers(0, m, size=count).astype(np.int32)
    probs["map_index"] = idx
    probs["cur_xy"] += offsets[idx]
    probs["goal_xyz"][:, :2] += offsets[idx]
    st, warm = synthetic.make_states(probs, 3)
    st_d, warm_d = st.copy(), warm.copy()
    want_c = np.zeros(count, dtype=abi.COMMAND_DTYPE)
    want_x = np.zeros((count, 9))
    for k in range(m):
        sel = np.where(idx == k)[0]
        st_k, warm_k = synthetic.make_states(probs[sel], 3)
        cc, xc, _ = _mirror(params, (cells[k], res, origins[k, 0], origins[k, 1]), probs[sel], st_k, warm_k)
        want_c[sel], want_x[sel] = cc, xc
    with BatchSolver(params) as s:
        s.set_costmap_pool(cells, res, origins)
        cg, xg = s.solve(probs, st, warm)
        _close(cg, want_c, frac=0.97)
        assert (cg["cost"] <= want_c["cost"] + 1e-6).mean() >= 0.97
        # the maps differ: solving everything against map 0 must NOT reproduce the per-map answers
        wrong = probs.copy()
        wrong["map_index"] = 0
        st_w, warm_w = synthetic.make_states(wrong, 3)
        cw, _ = s.solve(wrong, st_w, warm_w)
        other = idx != 0
        assert (np.abs(cw["cost"][other] - want_c["cost"][other]) > 1e-6).mean() > 0.3
        # out-of-range indices are clamped to the last map
        far = probs[idx == m - 1].copy()
        far["map_index"] = 1000
        st_f, warm_f = synthetic.make_states(far, 3)
        cf, _ = s.solve(far, st_f, warm_f)
        assert np.allclose(cf["vel"], cg["vel"][idx == m - 1], atol=1e-12)
        # device-resident pool, origins rewritten in place between ticks (rolling windows)
        dev = "cuda:0"
        d_cells = torch.from_numpy(cells).to(dev)
        d_orig = torch.from_numpy(origins).to(dev)
        s.set_costmap_pool(d_cells, res, d_orig)
        b = DeviceBatch(probs, st_d, warm_d, dev)
        s.solve_device(b.problems, b.states, b.warm, b.commands, solution=b.solution)
        torch.cuda.synchronize()
        cd = b.commands_host()
        assert np.allclose(cd["vel"], cg["vel"], atol=1e-12) and (cd["iterations"] == cg["iterations"]).all()


def test_device_entry_point_can_be_captured_in_a_hip_graph():
    """`neo_mpc_solve_batch_device` only enqueues work on the caller's stream (no allocation, no
    synchronisation), so a control tick can be captured once in a HIP graph and replayed: the replays,
    fed through the same device buffers, equal direct launches."""
    import torch
    from neo_mpc_planner2_amd.solver import BatchSolver, DeviceBatch
    params = util.orc.make_params()
    cmap = synthetic.make_costmap(300, seed=91)
    probs = synthetic.make_problems(256, 300, seed=92)
    st, warm = synthetic.make_states(probs, 3)
    dev = "cuda:0"
    with BatchSolver(params) as s:
        s.set_costmap(torch.from_numpy(cmap[0]).to(dev), *cmap[1:])
        direct = DeviceBatch(probs, st, warm, dev)
        graphed = DeviceBatch(probs, st, warm, dev)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):      # warm-up on the capture stream, as torch asks for
            s.solve_device(graphed.problems, graphed.states, graphed.warm, graphed.commands, solution=graphed.solution)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        graphed = DeviceBatch(probs, st, warm, dev)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            s.solve_device(graphed.problems, graphed.states, graphed.warm, graphed.commands, solution=graphed.solution)
        for tick in range(3):              # three ticks: state and warm start carry over inside the buffers
            g.replay()
            s.solve_device(direct.problems, direct.states, direct.warm, direct.commands, solution=direct.solution)
            torch.cuda.synchronize()
            a, b = graphed.commands_host(), direct.commands_host()
            assert (a["iterations"] == b["iterations"]).all() and (a["vel"] == b["vel"]).all(), tick
        assert (graphed.states_host()["has_old_goal"] == 1).all()


def test_pool_larger_than_one_ingest_launch_and_bad_footprints_are_refused():
    """K3 takes the map index from the grid's y coordinate: a pool of more than 65 535 maps is refused with
    NEO_MPC_ERR_UNSUPPORTED (not a generic launch failure); a non-finite footprint vertex is refused on the host
    (its rasterisation would walk billions of cells)."""
    import ctypes as C
    from neo_mpc_planner2_amd import _lib
    from neo_mpc_planner2_amd.solver import BatchSolver
    lib = _lib.load()
    with BatchSolver(util.orc.make_params()) as s:
        cells = np.zeros((2, 8, 8), dtype=np.uint8)
        orig = np.zeros((70000, 2))
        rc = lib.neo_mpc_set_costmap_pool(s._handle, C.c_void_p(cells.ctypes.data), 70000, 8, 8, 0.05,
                                          C.c_void_p(orig.ctypes.data))
        assert rc == -5 and b"65535" in lib.neo_mpc_last_error()
        cmap = synthetic.make_costmap(200, seed=1)
        s.set_costmap(*cmap)
        probs = synthetic.make_problems(100, 200, seed=2)
        st, warm = synthetic.make_states(probs, 3)
        fps = np.array([synthetic.footprint_world(r) for r in probs])
        fps[57, 2, 1] = np.inf
        with pytest.raises(_lib.NeoMpcError) as e:
            s.solve(probs, st, warm, footprints=fps)
        assert e.value.code == -1 and "not finite" in str(e.value)
        # a vertex far outside the map is fine: the outline's cost is decided (lethal) without walking to it
        fps[57, 2, 1] = 1e12
        cmds, _ = s.solve(probs, st, warm, footprints=fps)
        assert st["collision_footprint"][57] == 1 and (cmds["vel"][57] == 0.0).all()


def test_host_entry_point_on_page_locked_buffers():
    """neo_mpc_solve_batch queues its transfers on the null stream and waits once: on page-locked request /
    result buffers (a server's own arena) they are DMA transfers; the results are those of the pageable call."""
    import torch
    from neo_mpc_planner2_amd.solver import BatchSolver

    def pinned(a):
        t = torch.empty(a.nbytes, dtype=torch.uint8).pin_memory()
        v = t.numpy().view(a.dtype).reshape(a.shape)
        v[...] = a
        return v
    n = 3
    params = util.orc.make_params(control_steps=n)
    cfg, cmap, probs, st, warm = synthetic.make_workload("C2", seed=4, batch=1000)
    with BatchSolver(params) as s:
        s.set_costmap(*cmap)
        st_a, warm_a = st.copy(), warm.copy()
        cmds_a, x_a = s.solve(probs, st_a, warm_a)
        p_probs, p_st, p_warm = pinned(np.ascontiguousarray(probs)), pinned(st), pinned(warm)
        out = (pinned(np.zeros(len(probs), dtype=abi.COMMAND_DTYPE)), pinned(np.zeros((len(probs), 3 * n))))
        cmds_b, x_b = s.solve(p_probs, p_st, p_warm, out=out)
        assert cmds_b is out[0] and x_b is out[1]
        assert np.array_equal(x_a, x_b) and cmds_a.tobytes() == cmds_b.tobytes()
        assert st_a.tobytes() == p_st.tobytes() and np.array_equal(warm_a, p_warm)


@pytest.mark.parametrize("count", [4096, 2048, 1500, 5000])
def test_balanced_dispatch_changes_no_result(count):
    """neo_mpc_balance_dispatch_device: the dispatch order of the next device solves is rebuilt from the previous tick's
    iteration counts (K5) so that the searches sharing a SIMD need about the same number of iterations in total.  Instances
    are independent: commands, states, warm starts and solutions of every instance are bit for bit what they are in launch
    order -- which also says that the order is a permutation (an instance left out would keep its zeros, one solved twice
    would have advanced its state twice) -- for a one-round launch, a smaller multiple of 1024, and counts that get launch
    order (not a multiple of 1024; more than one round).  A batch of another size ignores the order; NULL disarms it."""
    import torch
    from neo_mpc_planner2_amd.solver import BatchSolver, DeviceBatch
    params = util.orc.make_params()
    cfg, cmap, probs, st, warm = synthetic.make_workload("C2", seed=5, batch=count)
    with BatchSolver(params) as s:
        s.set_costmap(*cmap)
        runs = []
        for balanced in (False, True):
            db = DeviceBatch(probs, st, warm, "cuda:0")
            P = db.problems.view(torch.float64).reshape(count, -1)
            per_tick = []
            for t in range(4):
                s.solve_device(db.problems, db.states, db.warm, db.commands, solution=db.solution)
                if balanced:
                    s.balance_dispatch(db.commands)
                P[:, 19:22] = db.commands.view(torch.float64).reshape(count, -1)[:, 0:3]     # cur_vel <- the command
                torch.cuda.synchronize()
                per_tick.append((db.commands_host().copy(), db.states_host().copy(), db.warm.cpu().numpy().copy(),
                                 db.solution.cpu().numpy().copy()))
            if balanced:   # a batch of another size on the same handle: launch order, same results as ever
                half = DeviceBatch(probs[: count // 2], st[: count // 2], warm[: count // 2], "cuda:0")
                s.solve_device(half.problems, half.states, half.warm, half.commands)
                torch.cuda.synchronize()
                assert half.commands_host().tobytes() == runs[0][0][0][: count // 2].tobytes()
                s.balance_dispatch(None)
            runs.append(per_tick)
    for t in range(4):
        for a, b in zip(runs[0][t], runs[1][t]):
            assert a.tobytes() == b.tobytes(), t
    it = runs[0][3][0]["iterations"]
    assert (it > 0).all() and it.max() > it.min()      # (a spread of loads: there was something to balance)
