"""Carrot (look-ahead) selection -- SURVEY §8f row 2: the step before the solver
(src/NeoMpcPlanner.cpp:83-104 plan pruning, 157-171 look-ahead distance, 173-189 look-ahead
point, 221-232 slow_down_ state machine).  The C++ plugin cannot be built here (nav2/tf2 absent),
so the oracle restatement (oracle/mpc_oracle.c part 3) is checked against an independent,
literal NumPy transcription of those lines, and the HIP kernel against the oracle."""
import math

import numpy as np
import pytest

from neo_mpc_planner2_amd import abi, synthetic
from oracle import c_oracle

LP = dict(lookahead_dist_min=0.4, lookahead_dist_max=0.8, lookahead_dist_close_to_goal=0.3,
          max_transform_dist=5.0)


def transcription(poses, robot, slow, fcost, lp):
    """cpp:66-135, 157-189, 221-232 line by line for one robot (planar transform)."""
    if len(poses) == 0:
        return dict(status=1)
    d = np.hypot(poses[:, 0] - robot[0], poses[:, 1] - robot[1])
    begin = int(np.argmin(d))                                   # min_by: first minimum
    closer = bool(d[-1] <= lp["lookahead_dist_close_to_goal"])
    far = np.nonzero(d[begin:] > lp["max_transform_dist"])[0]
    end = begin + int(far[0]) if len(far) else len(poses)
    if end == begin:
        return dict(status=2, begin=begin, end=end, closer=closer)
    la = lp["lookahead_dist_min"]
    if (not slow) or closer:
        la = lp["lookahead_dist_max"]
        if closer:
            la = lp["lookahead_dist_close_to_goal"]
    c, s = math.cos(robot[2]), math.sin(robot[2])
    dx, dy = poses[begin:end, 0] - robot[0], poses[begin:end, 1] - robot[1]
    lx, ly = c * dx + s * dy, -s * dx + c * dy
    hit = np.nonzero(np.hypot(lx, ly) >= la)[0]
    k = int(hit[0]) if len(hit) else end - begin - 1
    yaw = poses[begin + k, 2] - robot[2]
    yaw_w = math.atan2(math.sin(yaw), math.cos(yaw))            # what getRPY of the quaternion returns
    if abs(yaw_w) < 1.0:
        new_slow = 0
    elif abs(yaw_w) >= 1.0 and fcost > 200:
        new_slow = 1
    else:
        new_slow = 0
    # cpp:234-236: `if (footprint_cost == 255) throw ...` -- after the slow_down_ update, before the optimizer request
    return dict(status=3 if fcost == 255 else 0, begin=begin, end=end, closer=closer, la=la, xy=(lx[k], ly[k]), yaw=yaw_w,
                slow=new_slow)


def _inputs(count, seed):
    poses, offsets, robots = synthetic.make_plans(count, seed=seed, min_len=8, max_len=300)
    rng = np.random.default_rng(seed + 1)
    slow = rng.integers(0, 2, size=count).astype(np.int32)
    fcost = rng.choice([0.0, 150.0, 201.0, 253.0, 255.0], size=count)
    return poses, offsets, robots, slow, fcost


def test_oracle_restatement_matches_literal_transcription():
    poses, offsets, robots, slow, fcost = _inputs(300, seed=5)
    # edge cases: robot far from everything (status 2), robot at the goal (closer_to_goal)
    robots[0, :2] += 100.0
    robots[1, :2] = poses[offsets[2] - 1, :2]
    slow_in = slow.copy()
    car = c_oracle.select_carrots(poses, offsets, robots, slow, fcost, **LP)
    assert car["status"][0] == 2 and car["closer_to_goal"][1] == 1
    for i in range(len(robots)):
        t = transcription(poses[offsets[i]:offsets[i + 1]], robots[i], slow_in[i], fcost[i], LP)
        assert car["status"][i] == t["status"]
        if t["status"] in (1, 2):
            continue
        assert (car["begin"][i], car["end"][i]) == (t["begin"], t["end"])
        assert bool(car["closer_to_goal"][i]) == t["closer"] and car["lookahead_dist"][i] == t["la"]
        assert np.allclose(car["xy"][i], t["xy"], rtol=0, atol=1e-12)
        yaw = math.atan2(2 * car["q"][i][3] * car["q"][i][2], 1 - 2 * car["q"][i][2] ** 2)
        assert abs(yaw - t["yaw"]) <= 1e-12
        assert car["slow_down"][i] == t["slow"] == slow[i]


def test_oracle_empty_plan_and_problem_write():
    poses, offsets, robots, slow, fcost = _inputs(4, seed=9)
    offsets2 = np.concatenate([offsets[:2], offsets[1:]]).astype(np.uint32)   # robot 1 gets an empty plan
    robots2 = np.concatenate([robots[:1], robots[:1], robots[1:]])
    slow2 = np.ones(5, dtype=np.int32)
    probs = synthetic.make_problems(5, 200, seed=1)
    before = probs["carrot_xy"].copy()
    car = c_oracle.select_carrots(poses, offsets2, robots2, slow2, None, problems=probs, **LP)
    assert car["status"][1] == 1 and (probs["carrot_xy"][1] == before[1]).all()
    ok = car["status"] == 0
    assert (probs["carrot_xy"][ok] == car["xy"][ok]).all() and (probs["carrot_q"][ok] == car["q"][ok]).all()
    # every throw in front of the service call (cpp:70, 131, 235) means "no request this tick": the record says so
    assert (probs["skip"] == (car["status"] != 0)).all() and (probs["switch_opt"][ok] == car["closer_to_goal"][ok]).all()


def test_footprint_cost_255_makes_no_request():
    """cpp:234-236: a footprint cost of 255 throws AFTER the slow_down_ update and BEFORE the optimizer request: carrot
    status 3, the request record is marked `skip`, and the solver (here: the CPU mirror) leaves that robot's state, warm
    start and command alone."""
    from oracle import mpc_oracle as orc
    poses, offsets, robots, slow, fcost = _inputs(64, seed=31)
    fcost[::4] = 255.0
    probs = synthetic.make_problems(64, 500, seed=4)
    car = c_oracle.select_carrots(poses, offsets, robots, slow, fcost, problems=probs, **LP)
    thrown = fcost == 255.0
    assert ((car["status"] == 3) == (thrown & (car["status"] != 2) & (car["status"] != 1))).all() and (car["status"] == 3).any()
    assert (probs["skip"] == (car["status"] != 0)).all()
    cmap = synthetic.make_costmap(500, seed=0)
    st, warm = synthetic.make_states(probs, 3)
    warm[:] = 0.01
    st0, warm0 = st.copy(), warm.copy()
    cm, _, _ = c_oracle.solve_batch(orc.make_params(), cmap, probs, st, warm)
    sk = probs["skip"] != 0
    assert (cm["flags"][sk] == abi.FLAG_SKIPPED).all() and ((cm["flags"][~sk] & abi.FLAG_SKIPPED) == 0).all()
    assert st[sk].tobytes() == st0[sk].tobytes() and (warm[sk] == warm0[sk]).all() and (cm["vel"][sk] == 0.0).all()
    assert (cm["iterations"][sk] == 0).all() and (cm["cost"][sk] == 0.0).all()
    assert st[~sk].tobytes() != st0[~sk].tobytes()


@pytest.mark.gpu
def test_carrot_kernel_matches_oracle():
    from neo_mpc_planner2_amd.solver import BatchSolver
    poses, offsets, robots, slow, fcost = _inputs(4096, seed=11)
    robots[0, :2] += 100.0
    slow_g, slow_c = slow.copy(), slow.copy()
    pg = synthetic.make_problems(4096, 200, seed=2)
    pc = pg.copy()
    want = c_oracle.select_carrots(poses, offsets, robots, slow_c, fcost, problems=pc, **LP)
    with BatchSolver({}) as s:
        got = s.select_carrots(poses, offsets, robots, slow_g, fcost, problems=pg, **LP)
    for f in ("begin", "end", "closer_to_goal", "slow_down", "status", "lookahead_dist"):
        assert (got[f] == want[f]).all(), f
    assert (slow_g == slow_c).all()
    assert np.allclose(got["xy"], want["xy"], rtol=0, atol=1e-12)
    assert np.allclose(got["q"], want["q"], rtol=0, atol=1e-12)
    assert np.allclose(pg["carrot_xy"], pc["carrot_xy"], rtol=0, atol=1e-12)
    assert np.allclose(pg["carrot_q"], pc["carrot_q"], rtol=0, atol=1e-12)
    assert (got["status"] == 0).sum() >= 3000 and (got["status"] == 3).sum() >= 500
    assert (pg["skip"] == pc["skip"]).all() and (pg["switch_opt"] == pc["switch_opt"]).all() and (pg["skip"] != 0).sum() >= 500


@pytest.mark.gpu
def test_carrots_feed_the_solver_on_device():
    """device-resident chain K4 -> K1: carrots are written straight into the request records."""
    import torch
    from neo_mpc_planner2_amd.solver import BatchSolver, DeviceBatch
    from oracle import mpc_oracle as orc
    count = 1024
    poses, offsets, robots, slow, fcost = _inputs(count, seed=21)
    cmap = synthetic.make_costmap(500, seed=0)
    probs = synthetic.make_problems(count, 500, seed=3)
    st, warm = synthetic.make_states(probs, 3)
    # reference chain on the host: oracle carrots, then the GPU solver through host buffers
    pc, sc = probs.copy(), slow.copy()
    c_oracle.select_carrots(poses, offsets, robots, sc, fcost, problems=pc, **LP)
    with BatchSolver(orc.make_params()) as s:
        s.set_costmap(*cmap)
        want, _ = s.solve(pc, st.copy(), warm.copy())
        dev = "cuda:0"
        db = DeviceBatch(probs, st, warm, dev)
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
        d_poses, d_off, d_rob, d_slow, d_fc = t(poses), t(offsets.view(np.int32)), t(robots), t(slow), t(fcost)
        d_car = torch.zeros((count, abi.CARROT_DTYPE.itemsize), dtype=torch.uint8, device=dev)
        lp = abi.NeoMpcLookaheadParams(LP["lookahead_dist_min"], LP["lookahead_dist_max"],
                                       LP["lookahead_dist_close_to_goal"], LP["max_transform_dist"])
        s.select_carrots_device(lp, d_poses, d_off, d_rob, d_slow, d_car, d_fc, problems=db.problems)
        s.solve_device(db.problems, db.states, db.warm, db.commands)
        torch.cuda.synchronize()
        got = db.commands_host()
    dv = np.abs(got["vel"] - want["vel"]).max(axis=1)
    # the carrots agree to an ulp; the Newton path's finite-difference Hessian amplifies that a little
    assert (dv <= 1e-6).mean() >= 0.98 and (dv <= 1e-3).mean() >= 0.995


@pytest.mark.gpu
def test_skipped_instances_are_left_alone_by_the_solver():
    """neo_mpc_problem.skip (set by K4 with carrot status 3, cpp:234-236): K1 and K2 leave the robot's state record and warm
    start alone and answer zero twist + NEO_MPC_FLAG_SKIPPED (zero rows in the optional outputs) -- on the staged path, the
    small latency path and the in-place (page-locked) path; the other instances come out exactly as without any skip."""
    import ctypes as C
    from neo_mpc_planner2_amd import _lib
    from neo_mpc_planner2_amd.solver import BatchSolver
    from oracle import mpc_oracle as orc
    lib = _lib.load()
    cmap = synthetic.make_costmap(500, seed=0)
    for count in (48, 700):                      # latency path / staged path
        probs = synthetic.make_problems(count, 500, seed=6)
        st0, warm0 = synthetic.make_states(probs, 3)
        warm0[:] = 0.02
        with BatchSolver(orc.make_params()) as s:
            s.set_costmap(*cmap)
            st_a, warm_a = st0.copy(), warm0.copy()
            ref, xref = s.solve(probs, st_a, warm_a)
            sk = np.zeros(count, dtype=bool)
            sk[::3] = True
            p2 = probs.copy()
            p2["skip"] = sk
            st_b, warm_b = st0.copy(), warm0.copy()
            cmds = np.zeros(count, dtype=abi.COMMAND_DTYPE)
            cmds["vel"] = 7.0
            sol = np.full((count, 9), 5.0)
            s.solve(p2, st_b, warm_b, out=(cmds, sol))
            assert (cmds["flags"][sk] == abi.FLAG_SKIPPED).all() and (cmds["vel"][sk] == 0.0).all() and (sol[sk] == 0.0).all()
            assert (cmds["iterations"][sk] == 0).all() and (cmds["status"][sk] == 0).all() and (cmds["cost"][sk] == 0.0).all()
            assert st_b[sk].tobytes() == st0[sk].tobytes() and (warm_b[sk] == warm0[sk]).all()
            assert cmds[~sk].tobytes() == ref[~sk].tobytes() and (sol[~sk] == xref[~sk]).all()
            assert st_b[~sk].tobytes() == st_a[~sk].tobytes() and (warm_b[~sk] == warm_a[~sk]).all()
            # K2 alone
            st_c, warm_c = st0.copy(), warm0.copy()
            c2 = s.postprocess(p2, st_c, warm_c, xref)
            assert (c2["flags"][sk] == abi.FLAG_SKIPPED).all() and st_c[sk].tobytes() == st0[sk].tobytes()
            if count > 64:   # in place on page-locked arrays
                arrays = [np.ascontiguousarray(p2).copy(), st0.copy(), warm0.copy(), cmds.copy(), sol.copy()]
                arrays[3]["vel"] = 7.0
                arrays[3]["flags"] = 0
                arrays[4][:] = 5.0
                for a in arrays:
                    assert lib.neo_mpc_pin_host_memory(C.c_void_p(a.ctypes.data), a.nbytes) == 0
                try:
                    b = abi.batch_struct(*arrays)
                    _lib.check(lib.neo_mpc_solve_batch(s._handle, C.byref(b)))
                    assert arrays[3].tobytes() == cmds.tobytes() and (arrays[4] == sol).all()
                    assert arrays[1].tobytes() == st_b.tobytes() and (arrays[2] == warm_b).all()
                finally:
                    for a in arrays:
                        assert lib.neo_mpc_unpin_host_memory(C.c_void_p(a.ctypes.data)) == 0


@pytest.mark.gpu
def test_skip_fields_other_than_0_or_1_are_refused_on_host_batches():
    """neo_mpc_problem.skip was reserved[0] in ABI 1 and that header never asked for zeroed reserved bytes: garbage there
    must not take robots out of a tick silently.  Host batches are scanned (NEO_MPC_ERR_INVALID_ARGUMENT, nothing
    touched); the device entry points act on skip == 1 alone -- any other value is solved like 0."""
    import torch
    from neo_mpc_planner2_amd import _lib
    from neo_mpc_planner2_amd.solver import BatchSolver, DeviceBatch
    from oracle import mpc_oracle as orc
    cmap = synthetic.make_costmap(500, seed=0)
    probs = synthetic.make_problems(96, 500, seed=8)
    st0, warm0 = synthetic.make_states(probs, 3)
    with BatchSolver(orc.make_params()) as s:
        s.set_costmap(*cmap)
        ref, xref = s.solve(probs, st0.copy(), warm0.copy())
        bad = probs.copy()
        bad["skip"][5] = 0x3f800000          # (what an uninitialised float 1.0 looks like in the old reserved slot)
        st, warm = st0.copy(), warm0.copy()
        with pytest.raises(_lib.NeoMpcError) as err:
            s.solve(bad, st, warm)
        assert err.value.code == -1 and "skip" in str(err.value)      # NEO_MPC_ERR_INVALID_ARGUMENT
        assert st.tobytes() == st0.tobytes() and (warm == warm0).all()
        # device batch: the value is not 1, so the robot is solved
        db = DeviceBatch(bad, st0.copy(), warm0.copy(), "cuda:0")
        s.solve_device(db.problems, db.states, db.warm, db.commands, db.solution)
        torch.cuda.synchronize()
        got = db.commands_host()
        assert (got["flags"] & abi.FLAG_SKIPPED == 0).all() and got.tobytes() == ref.tobytes()
