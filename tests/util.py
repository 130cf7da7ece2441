"""Shared helpers for the parity tests: fixture loading and record <-> oracle glue."""
import os

import numpy as np

from neo_mpc_planner2_amd.abi import PROBLEM_DTYPE
from oracle import mpc_oracle as orc

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def params_from(keys, vec):
    p = {str(k): float(v) for k, v in zip(keys, vec)}
    p["control_steps"] = int(p["control_steps"])
    return p


def problems_from(raw):
    return np.ascontiguousarray(raw).view(PROBLEM_DTYPE).reshape(raw.shape[:-1])


def oracle_problem(row, footprint=None):
    fp = ()
    if footprint is not None and not np.isnan(footprint).any():
        fp = [tuple(p) for p in footprint]
    return orc.Problem(row["cur_xy"], row["cur_q"], row["carrot_xy"], row["carrot_q"],
                       row["goal_xyz"], row["goal_q"], row["cur_vel"],
                       control_interval=float(row["control_interval"]),
                       delta_t=float(row["delta_t"]), footprint=fp)


def oracle_costmap(cells, meta):
    return orc.Costmap(cells, meta[0], meta[1], meta[2])



def load_seam_dump(path):
    """`examples/plugin_seam.cpp --run --dump FILE`: header (tick count, control_steps, map size, parameters, map
    geometry, the costmap versions) and one record per control tick -- what the C++ seam sent, held and got."""
    import ctypes as C
    from neo_mpc_planner2_amd import abi
    raw = open(path, "rb").read()
    assert raw[:8] == b"NEOSEAM1", raw[:8]
    ticks, n_steps, size, n_maps = np.frombuffer(raw, dtype="<i4", count=4, offset=8)
    off = 24
    ps = abi.NeoMpcParams.from_buffer_copy(raw[off:off + C.sizeof(abi.NeoMpcParams)])
    off += C.sizeof(abi.NeoMpcParams)
    params = {name: getattr(ps, name) for name, _ in abi.NeoMpcParams._fields_ if name != "reserved_i"}
    geom = np.frombuffer(raw, dtype="<f8", count=3, offset=off)
    off += 24
    maps = np.frombuffer(raw, dtype=np.uint8, count=n_maps * size * size, offset=off).reshape(n_maps, size, size)
    off += n_maps * size * size
    nv = 3 * int(n_steps)
    rec = np.dtype([("map", "<i4"), ("tick", "<i4"), ("problem", abi.PROBLEM_DTYPE), ("state_before", abi.STATE_DTYPE),
                    ("warm_before", "<f8", (nv,)), ("command", abi.COMMAND_DTYPE), ("state_after", abi.STATE_DTYPE),
                    ("warm_after", "<f8", (nv,))])
    assert len(raw) - off == int(ticks) * rec.itemsize, (len(raw) - off, int(ticks), rec.itemsize)
    return params, tuple(geom), maps, np.frombuffer(raw, dtype=rec, count=int(ticks), offset=off)



def solve_group(name, prefix):
    """One `_g3_group` of a cold-solve fixture (oracle/gen_golden.py): (arrays by bare name, params, problems,
    has_map mask)."""
    g = load(name)
    params = params_from(g["param_keys"], g[prefix + "params"])
    group = {k[len(prefix):]: g[k] for k in g.files if k.startswith(prefix)} if prefix else {k: g[k] for k in g.files}
    probs = problems_from(group["problems"])
    hm = group["has_map"].astype(bool) if "has_map" in group else np.zeros(len(probs), dtype=bool)
    return group, params, probs, hm


#: (fixture, prefix) of every cold-solve group made at parameters other than the README's at opt_tolerance 1e-3
G9_GROUPS = (("g9_solves_pydefaults.npz", "n3_"), ("g9_solves_pydefaults.npz", "n8_"))
G8_MID_GROUPS = tuple(("g8_mid.npz", "r%02d_" % r) for r in (10, 15, 20, 25, 30))
#: G10 (round 4): held-out parameter sets nobody looked at while the solver's thresholds were tuned -- (set, control_steps)
G10_GROUPS = tuple((name, n) for name in "abc" for n in (3, 5, 8, 12))
#: G11 (round 4): optimizer() episodes of the reference run to convergence (opt_tolerance 1e-12) on an all-free map
G11_FIXTURES = ("g11_warm_converged.npz", "g11_warm_converged_n8.npz")
#: G12 / G13 (round 4, drawn AFTER the last change to the search or a threshold): G10's protocol at three more parameter sets
#: and control_steps 3 / 4 / 6 / 10; G11's protocol at G10's set "a" (heavy control weight, box cutting the disc)
G12_GROUPS = tuple((name, n) for name in "def" for n in (3, 4, 6, 10))
G13_FIXTURES = ("g13_warm_converged_set_a.npz", "g13_warm_converged_set_a_n5.npz")
EPISODE_FIXTURES = ("g4_episodes.npz", "g4_episodes_n8.npz", "g4_episodes_params.npz", "g9_episodes_pydefaults.npz")



def closed_loop_on_the_mirror(params, cmap, probs, ticks, hz=30.0):
    """The fleet loop of neo_mpc_planner2_amd/fleet.py on the CPU mirror (oracle/mpc_oracle.c part 2): robots moved by
    their own commands, the carrot kept on its initial world bearing, warm start = the reference's shift.  Returns the
    per-tick command records."""
    from neo_mpc_planner2_amd import synthetic
    from oracle import c_oracle

    def yaw_of(q):
        return np.arctan2(2 * (q[:, 3] * q[:, 2] + q[:, 0] * q[:, 1]), 1 - 2 * (q[:, 1] ** 2 + q[:, 2] ** 2))
    probs = probs.copy()
    st, warm = synthetic.make_states(probs, params["control_steps"])
    yaw, pos = yaw_of(probs["cur_q"]).copy(), probs["cur_xy"].copy()
    c, s = np.cos(yaw), np.sin(yaw)
    off = np.stack([c * probs["carrot_xy"][:, 0] - s * probs["carrot_xy"][:, 1],
                    s * probs["carrot_xy"][:, 0] + c * probs["carrot_xy"][:, 1]], 1)
    carrot_yaw_w = yaw + yaw_of(probs["carrot_q"])
    probs["control_interval"] = 1.0 / hz
    probs["delta_t"] = 1.0 / hz
    out = []
    for _ in range(ticks):
        cm = c_oracle.solve_batch(params, cmap, probs, st, warm)[0]
        v = cm["vel"]
        yaw = yaw + v[:, 2] / hz
        c, s = np.cos(yaw), np.sin(yaw)
        pos = pos + np.stack([c * v[:, 0] - s * v[:, 1], s * v[:, 0] + c * v[:, 1]], 1) / hz
        probs["cur_xy"], probs["cur_q"] = pos, synthetic.yaw_quat(yaw)
        probs["carrot_xy"][:, 0] = c * off[:, 0] + s * off[:, 1]
        probs["carrot_xy"][:, 1] = -s * off[:, 0] + c * off[:, 1]
        probs["carrot_q"] = synthetic.yaw_quat(carrot_yaw_w - yaw)
        probs["cur_vel"] = v
        out.append(cm)
    return out


orc = orc  # re-export: tests use util.orc.make_params


#: how many all-free-map cases of a held-out group have to count for the first-control gate: 16 of the 24 (round 4's bar),
#: except where the REFERENCE's own sixteen answers leave fewer unique -- set "a" (heavy control weight: SLSQP at ftol 1e-12
#: stops up to 1.7e-2 apart in a valley that flat) and "d" at control_steps 10: there the fixture's own count is the bar
MIN_UNIQUE = {("a", 5): 15, ("a", 8): 10, ("a", 12): 9, ("d", 10): 12}


def check_held_out_group(solve, name, n_steps, p2_bar=3e-4, fixture="g10_heldout.npz", min_ok=None):
    """G10 gates for one (set, control_steps) group.  `solve(params, cmap, problems) -> (commands, x)` is the build's cold
    solve (GPU through the C-ABI, or the CPU mirror).  P3 on every case: f <= f(SLSQP as shipped, ftol = the set's
    opt_tolerance) + 1e-3, feasible, converged.  P2 on the all-free-map cases the FIXTURE flags `unique` -- the reference's
    SLSQP at ftol 1e-12 reports status 0 and ends within 1e-4 of the same first control from fifteen other starts
    (gen_golden._g3_group): |u0 - u0(SLSQP 1e-12)|_inf <= p2_bar (the north star's bar is 1e-3; the round-3 judge asked
    for 3e-4 of margin on sets that were never tuned on).  Which cases count is decided by the reference's answers alone,
    never by the build's objective value; the cases that are not unique (more than one KKT point, or a run of the
    reference that stalled short) are reported, not gated.  Returns the margins for the report line."""
    grp, params, probs, hm = solve_group(fixture, "%s_n%d_" % (name, n_steps))
    assert params["control_steps"] == n_steps and len(probs) >= 48
    out = {}
    for tag, mask, cells in (("free", ~hm, np.zeros_like(grp["cells"])), ("map", hm, grp["cells"])):
        cmap = (cells,) + tuple(grp["map_meta"])
        cmds, x = solve(params, cmap, probs[mask])
        worse = cmds["cost"] - grp["f_loose"][mask]
        assert (worse <= 1e-3).all(), (name, n_steps, tag, worse.max())                                    # P3
        assert (cmds["status"] == 0).all()
        xs = x.reshape(len(x), -1, 3)
        assert (np.hypot(xs[:, :, 0], xs[:, :, 1]) <= params["max_vel_trans"] + 1e-9).all()
        for q, axis in enumerate(("x", "y", "theta")):
            assert (xs[:, :, q] <= params["max_vel_" + axis] + 1e-12).all() and (xs[:, :, q] >= params["min_vel_" + axis] - 1e-12).all()
        out["p3_" + tag] = worse.max()
        if tag == "free":
            unique = grp["unique"][mask].astype(bool)
            others = (grp["status_tight"][mask] == 0) & ~unique
            assert unique.sum() >= (MIN_UNIQUE.get((name, n_steps), 16) if min_ok is None else min_ok), (name, n_steps, int(unique.sum()))
            du0 = np.abs(x[:, :3] - grp["x_tight"][mask][:, :3]).max(axis=1)
            assert du0[unique].max() <= p2_bar, (name, n_steps, du0[unique].max())                          # P2
            assert (cmds["cost"] <= grp["f_tight"][mask] + 1e-4).all()
            out["p2"] = du0[unique].max()
            out["p2_cases"] = int(unique.sum())
            out["not_unique"] = int(others.sum())
            out["p2_not_unique"] = float(du0[others].max()) if others.any() else 0.0
        out["it_" + tag] = cmds["iterations"].mean()
    return out


def warm_gate(solve, postprocess, fixture):
    """G11 / G13: every call of the reference's CONVERGED episodes (SLSQP at ftol 1e-12, maxiter 500, all-free map) solved by
    the build at the README tolerance from the reference's own state -- its warm start, last_control, goal bookkeeping; the
    states are advanced with the reference's raw x.x injected (P5), so call k starts exactly where the reference did.
    `solve(params, cmap, rows, states, warm) -> (commands, x)`, `postprocess(params, cmap, rows, states, warm, x, success)`.
    Returns |command - reference command|_inf and |u0 - reference u0|_inf over the calls the reference converged on
    (status 0), the iteration counts, and the fixture's `settled` flag of those calls (the reference's own answer, the same
    solve taken up again from that answer and one from zeros agree on the first control to 1e-4: gen_golden.gen_g4)."""
    from neo_mpc_planner2_amd import abi
    g = load(fixture)
    params = params_from(g["param_keys"], g["params"])
    assert params["opt_tolerance"] == 1e-12 and not g["cells"].any()
    params["opt_tolerance"] = 1e-3        # the build runs at the shipped tolerance; the reference was run to convergence
    n = params["control_steps"]
    cmap = (g["cells"],) + tuple(g["map_meta"])
    probs = problems_from(g["problems"])
    n_ep, n_calls = probs.shape
    states, warm = abi.new_states(n_ep, n)
    dv, du, its = [], [], []
    for k in range(n_calls):
        rows = probs[:, k].copy()
        rows["footprint_cost"] = 0.0
        cmds, x = solve(params, cmap, rows, states.copy(), warm.copy())
        dv.append(np.abs(cmds["vel"] - g["out"][:, k]).max(axis=1))
        du.append(np.abs(x[:, :3] - g["raw_x"][:, k][:, :3]).max(axis=1))
        its.append(cmds["iterations"].copy())
        assert (cmds["status"] == 0).all()
        postprocess(params, cmap, rows, states, warm, g["raw_x"][:, k], g["success"][:, k])   # the reference's next state
        assert np.allclose(states["last_control"], g["last_control"][:, k], rtol=0, atol=1e-13)
    ok = g["success"].astype(bool)
    dv, du, its = np.array(dv).T, np.array(du).T, np.array(its).T
    return dv[ok], du[ok], its, g["settled"].astype(bool)[ok]


def p3w_group(solve, postprocess, params, cmap, grp):
    """P3w on one set of recorded episodes (`grp`: gen_golden.gen_g4's arrays): every call solved by the build from the
    REFERENCE's own state on the real costmap; the states are advanced with the reference's raw x.x injected.  Returns
    f(build) - f(reference's raw x.x) per (episode, call), the iteration counts and how many searches ran into the iteration cap.
    `solve(params, cmap, rows, states, warm) -> (commands, x)`, `postprocess(params, cmap, rows, states, warm, x, success)`."""
    from neo_mpc_planner2_amd import abi
    from oracle import c_oracle
    n = params["control_steps"]
    probs = problems_from(grp["problems"])
    n_ep, n_calls = probs.shape
    states, warm = abi.new_states(n_ep, n)
    worse, its, capped = [], [], 0
    for k in range(n_calls):
        fp = grp["footprint"][:, k]
        rows = probs[:, k].copy()
        has = ~np.isnan(fp).any(axis=(1, 2))
        rows["footprint_cost"] = 0.0
        if has.any():
            rows["footprint_cost"][has] = c_oracle.footprint_cost_batch(cmap, fp[has])
        cm, x = solve(params, cmap, rows, states.copy(), warm.copy())
        capped += int((cm["status"] != 0).sum())
        worse.append(cm["cost"] - c_oracle.objective_batch(params, cmap, rows, grp["raw_x"][:, k]))
        its.append(cm["iterations"].copy())
        postprocess(params, cmap, rows, states, warm, grp["raw_x"][:, k], grp["success"][:, k])
        assert np.allclose(states["last_control"], grp["last_control"][:, k], rtol=0, atol=1e-13)
    return np.array(worse).T, np.array(its).T, capped


def p3w_random_sets(solve, postprocess, fixture="g17_warm_costmap_sets.npz"):
    """G17: P3w at RANDOM parameter sets on the costmap (oracle/gen_golden.py gen_g17).  Returns per set (seed, control_steps,
    largest f(build) - f(reference), calls more than 1e-3 above, calls, mean iterations)."""
    g = load(fixture)
    cmap = (g["cells"],) + tuple(g["map_meta"])
    rows = []
    for seed, n in zip(g["seeds"], g["steps"]):
        pre = "s%d_" % seed
        grp = {k[len(pre):]: g[k] for k in g.files if k.startswith(pre)}
        params = params_from(g["param_keys"], grp["params"])
        assert params["control_steps"] == n
        worse, its, capped = p3w_group(solve, postprocess, params, cmap, grp)
        assert capped == 0, seed
        rows.append((int(seed), int(n), float(worse.max()), int((worse > 1e-3).sum()), int(worse.size), float(its.mean())))
    return rows


def assert_warm_gate(dv, settled, fixture):
    """The gate on warm_gate()'s command differences: >= 99.9 % of the SETTLED ticks within 1e-3 of the reference's
    converged command (which ticks are settled is the fixture's statement about the reference's own answers -- the build's
    objective value decides nothing), >= 99 % of all ticks, none beyond 5e-3."""
    assert settled.mean() >= 0.9, (fixture, settled.mean())
    at = dv[settled]
    assert (at <= 1e-3).mean() >= 0.999, (fixture, (at > 1e-3).sum(), at.size, at.max())
    assert (dv <= 1e-3).mean() >= 0.99 and dv.max() <= 5e-3, (fixture, (dv > 1e-3).sum(), dv.size, dv.max())
    return int((dv > 1e-3).sum()), int((at > 1e-3).sum()), int((~settled).sum())


def warm_drift(solve, count=4096, ticks=12):
    """Closed-loop drift: `count` robots on an all-free map, pose fixed, velocity = the previous command, `ticks` warm ticks
    at the README tolerance; then the same state solved once more at the README tolerance and once run to the end (400
    iterations, tolerances 1e-9 / 1e-12, no window rules).  `solve(params, cmap, problems, states, warm) -> (commands, x)`
    updates states / warm in place.  Returns |first control difference|, |command difference|, objective difference,
    iteration counts."""
    from neo_mpc_planner2_amd import synthetic
    cmap = synthetic.make_costmap(500, seed=0)
    zero = (np.zeros_like(cmap[0]),) + tuple(cmap[1:])
    p = synthetic.make_problems(count, 500, seed=77)
    st, warm = synthetic.make_states(p, 3)
    params = orc.make_params()
    for _ in range(ticks):
        cm, _ = solve(params, zero, p, st, warm)
        p["cur_vel"] = cm["vel"]
    c1, x1 = solve(params, zero, p, st.copy(), warm.copy())
    tight = dict(params, window_tolerance=-1.0, step_tolerance=1e-9, cost_tolerance=1e-12, max_iterations=400)
    c2, x2 = solve(tight, zero, p, st.copy(), warm.copy())
    return (np.abs(x1[:, :3] - x2[:, :3]).max(axis=1), np.abs(c1["vel"] - c2["vel"]).max(axis=1), c1["cost"] - c2["cost"],
            c1["iterations"], c2["iterations"])


def assert_warm_drift(du, dv, df):
    """The gate (round 5): after 12 warm ticks no COMMAND is more than 1e-3 from the run-to-the-end kernel's (round 4: 4 of
    4096, max 3.7e-3), no first control more than 2e-3 and at most two beyond 1e-3 (round 4: 8, max 7.4e-3)."""
    assert (dv > 1e-3).sum() == 0, ((dv > 1e-3).sum(), dv.max())
    assert (du > 1e-3).sum() <= 2 and du.max() <= 2e-3, ((du > 1e-3).sum(), du.max())
    assert df.max() <= 1e-4


#: (round 4) two searches the random-parameter fuzz found stopping short; the values are the fuzz's draws
CORNER_SET = dict(control_steps=10, max_vel_x=0.8074363686083792, min_vel_x=-0.6529286462016756, max_vel_y=0.6821757373561252,
                  min_vel_y=-0.6821757373561252, max_vel_trans=0.8190297446276567, max_vel_theta=1.311122200647496,
                  min_vel_theta=-1.046923033825679, w_trans=1.851846062907643, w_orient=1.0654822176259133,
                  w_control=0.07955025485374233, w_terminal=0.014261579164782094, w_costmap=0.22656099959405637,
                  prediction_horizon=0.6553971944084013)
CLOSING_SET = dict(control_steps=3, max_vel_x=0.4687266244102684, min_vel_x=-0.13636412838390244, max_vel_y=0.23990841075303393,
                   min_vel_y=-0.23990841075303393, max_vel_trans=0.6346691128830954, max_vel_theta=1.0744619081196514,
                   min_vel_theta=-0.7455525260186087, w_trans=1.624981170062613, w_orient=0.9471912249885202,
                   w_control=0.35504224630224157, w_terminal=0.29155579279540883, w_costmap=0.24117871887188674,
                   prediction_horizon=0.7451387226351949)
CORNER3_SET = dict(control_steps=3, max_vel_x=0.896726222782037, min_vel_x=-0.526036149425443, max_vel_y=0.42580420912603656,
                   min_vel_y=-0.42580420912603656, max_vel_trans=0.98382188609014, max_vel_theta=1.5369900040891644,
                   min_vel_theta=-1.0561500863189393, w_trans=1.7833138660348593, w_orient=0.6860036735692177,
                   w_control=0.047025036947067854, w_terminal=0.052826917579506354, w_costmap=0.042441839011586456,
                   prediction_horizon=0.7327823311689736)


def check_stop_rule_regressions(solve):
    """`solve(params, cmap, problems) -> (commands, x)`, cold.  (a) One-sided slides: with the box cutting the disc, blocks
    sit in corners of the feasible set; a Newton step that sent one of them out of its corner was no descent direction at
    any length and the search crept on proximal steps (instance 233: 21 iterations, stopped 1.4e-2 short) -- such blocks
    are pinned and the direction computed once more.  (b) The closing-in rule fired on a quadratically converging Newton
    search followed by ONE blocked iteration (instance 2: 4.8e-3 short).  Both against the same search run to the end."""
    from neo_mpc_planner2_amd import synthetic
    out = {}
    for tag, pset, seed, inst, it_cap in (("corner", CORNER_SET, 110, 233, 15), ("closing", CLOSING_SET, 115, 2, 8),
                                          ("corner, dense direction", CORNER3_SET, 117, 151, 9)):   # (instance 151: 1.2e-2 short)
        params = orc.make_params(**pset)
        _, cmap, probs, _, _ = synthetic.make_workload("C2", seed=seed, batch=256)
        free = (np.zeros_like(cmap[0]),) + tuple(cmap[1:])
        tight = dict(params, window_tolerance=-1.0, step_tolerance=1e-10, cost_tolerance=1e-14, max_iterations=400)
        c1, x1 = solve(params, free, probs)
        c2, x2 = solve(tight, free, probs)
        du = np.abs(x1[:, :3] - x2[:, :3]).max(axis=1)
        assert du[inst] <= 1e-3 and c1["iterations"][inst] <= it_cap, (tag, du[inst], c1["iterations"][inst])
        assert (du > 1e-3).sum() <= 1 and du.max() <= 2.5e-3, (tag, np.sort(du)[-3:])
        assert (c1["cost"] <= c2["cost"] + 1e-5).all()
        out[tag] = (du[inst], du.max(), int(c1["iterations"][inst]))
    return out


def random_sets_miss_rates(solve, fixture="g14_random_sets.npz", only_steps=None, over=None):
    """G14 / G15: RANDOM parameter sets (oracle/fuzz_reference.py's draws) x 24 cold problems under G10's protocol.  Counts:
    P3 misses (objective more than 1e-3 above SLSQP as shipped) on all-free maps and on costmaps, cases where SLSQP as
    shipped is more than 1e-3 above the build, P2 misses (first control more than 1e-3 from SLSQP run to the end) over the
    cases the fixture flags `unique` (check_held_out_group), and -- reported only -- the largest distance over the status-0
    cases that are not.  `solve(params, cmap, problems) -> (commands, x)`, cold."""
    g = load(fixture)
    out = dict(cases_free=0, cases_map=0, p3_miss_free=0, p3_miss_map=0, p3_worst=-1.0, ref_worse=0, p2_cases=0, p2_miss=0, p2_worst=0.0,
               not_unique=0, p2_not_unique_worst=0.0, misses=[])
    for seed, n in zip(g["seeds"], g["steps"]):
        if only_steps is not None and n != only_steps:
            continue
        pre = "s%d_" % seed
        grp = {k[len(pre):]: g[k] for k in g.files if k.startswith(pre)}
        params = dict(params_from(g["param_keys"], grp["params"]), **(over or {}))
        assert params["control_steps"] == n
        probs = problems_from(grp["problems"])
        hm = grp["has_map"].astype(bool)
        for tag, mask, cells in (("free", ~hm, np.zeros_like(grp["cells"])), ("map", hm, grp["cells"])):
            cmds, x = solve(params, (cells,) + tuple(grp["map_meta"]), probs[mask])
            assert (cmds["status"] == 0).all()
            worse = cmds["cost"] - grp["f_loose"][mask]
            out["cases_" + tag] += int(mask.sum())
            out["p3_miss_" + tag] += int((worse > 1e-3).sum())
            out["p3_worst"] = max(out["p3_worst"], float(worse.max()))
            out["ref_worse"] += int((worse < -1e-3).sum())
            out["misses"] += [("P3", int(seed), int(np.nonzero(mask)[0][i]), float(worse[i])) for i in np.nonzero(worse > 1e-3)[0]]
            if tag == "free":
                unique = grp["unique"][mask].astype(bool)
                others = (grp["status_tight"][mask] == 0) & ~unique
                du0 = np.abs(x[:, :3] - grp["x_tight"][mask][:, :3]).max(axis=1)
                out["p2_cases"] += int(unique.sum())
                out["p2_miss"] += int((du0[unique] > 1e-3).sum())
                out["p2_worst"] = max(out["p2_worst"], float(du0[unique].max()) if unique.any() else 0.0)
                out["not_unique"] += int(others.sum())
                out["p2_not_unique_worst"] = max(out["p2_not_unique_worst"], float(du0[others].max()) if others.any() else 0.0)
                out["misses"] += [("P2", int(seed), int(np.nonzero(mask)[0][i]), float(du0[i])) for i in np.nonzero(unique & (du0 > 1e-3))[0]]
    return out


#: fixture -> (random parameter sets, all-free-map cases the reference's own answers flag unique at least)
RANDOM_SETS = {"g14_random_sets.npz": 48, "g15_judge_sets.npz": 64, "g16_judge_sets_r5.npz": 106,
               "g18_held_out_sets.npz": 48}


def assert_random_sets(m, fixture="g14_random_sets.npz"):
    """Exact gates (round 5): no P3 miss on any of the random sets' cases, all-free map or costmap; no P2 miss on the cases
    the reference's own answers flag unique."""
    assert m["cases_free"] == m["cases_map"] == 12 * RANDOM_SETS[fixture]
    assert m["p3_miss_free"] == 0 and m["p3_miss_map"] == 0, m
    assert m["p2_miss"] == 0 and m["p2_worst"] <= 1e-3 and m["p2_cases"] >= 0.8 * m["cases_free"], m
    assert m["ref_worse"] >= 10 * RANDOM_SETS[fixture], m     # (the reference at its shipped tolerance is the looser of the two by far)
