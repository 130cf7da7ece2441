#!/usr/bin/env python3
"""Parity report of the HIP solver against the reference's SciPy path (SURVEY.md §8c tiers P2-P4), printed as
text.  Runs on the GPU box; uses tests/golden/g3_solves.npz (SLSQP solves of the REFERENCE's objective at
ftol 1e-3 and 1e-12, cold start, control_steps 3 / 8 / 32) and the G4 episodes (the reference's own warm
starts on real costmaps).  Every distribution is given twice: on the raw first control u0 and on the COMMAND
the robot receives (after the low-pass and the acceleration clamp of py:366-367, 383-391, with last_control =
the current velocity) -- K2 applied to the build's x and, through the oracle's wrapper, to SLSQP's x.

    python tools/parity_report.py > profiles/rNN_parity_report.txt
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from neo_mpc_planner2_amd import abi, synthetic  # noqa: E402
from neo_mpc_planner2_amd.solver import BatchSolver  # noqa: E402
from oracle import c_oracle  # noqa: E402
from tests import util  # noqa: E402


def pct(a):
    return "median %.2e  p90 %.2e  p99 %.2e  max %.2e" % (np.median(a), np.percentile(a, 90), np.percentile(a, 99), a.max())


def command_of(params, cmap, pr, x):
    """the reference's post-filter (py:365-403) applied to a raw solver output, mid-episode state"""
    st, warm = synthetic.make_states(pr, params["control_steps"])
    cmds, _, _ = c_oracle.postprocess_batch(params, cmap, pr, st, warm, x.copy())
    return cmds["vel"]


def cold_starts():
    g = util.load("g3_solves.npz")
    print("# HIP solver vs SciPy SLSQP on the reference objective (%s)" % str(g["versions"]))
    for n in (3, 8, 32):
        k = "" if n == 3 else "n%d_" % n
        params = util.params_from(g["param_keys"], g[k + "params"])
        probs = util.problems_from(g[k + "problems"])
        hm = g[k + "has_map"].astype(bool)
        for name, mask, cells in (("zero costmap (unique minimiser)", ~hm, np.zeros_like(g[k + "cells"])),
                                  ("synthetic costmap (local minima)", hm, g[k + "cells"])):
            cmap = (cells,) + tuple(g[k + "map_meta"])
            pr = probs[mask]
            st, warm = synthetic.make_states(pr, n)
            with BatchSolver(params) as s:
                s.set_costmap(*cmap)
                cmds, x = s.solve(pr, st, warm)
            xt, xl = g[k + "x_tight"][mask], g[k + "x_loose"][mask]
            ok = g[k + "status_tight"][mask] == 0
            du_t = np.abs(x[:, :3] - xt[:, :3]).max(axis=1)
            du_l = np.abs(x[:, :3] - xl[:, :3]).max(axis=1)
            ref_ll = np.abs(xl[:, :3] - xt[:, :3]).max(axis=1)
            c_b, c_t, c_l = cmds["vel"], command_of(params, cmap, pr, xt), command_of(params, cmap, pr, xl)
            dc_t, dc_l, dc_ref = (np.abs(c_b - c_t).max(axis=1), np.abs(c_b - c_l).max(axis=1), np.abs(c_l - c_t).max(axis=1))
            print("\n## control_steps %d, %s, %d cold-start problems (SLSQP ftol=1e-12 reached status 0 on %d)"
                  % (n, name, mask.sum(), ok.sum()))
            print("P2  |u0 - u0(SLSQP ftol=1e-12)|_inf      : %s" % pct(du_t))
            print("    ... on the command                   : %s" % pct(dc_t))
            print("P3  f - f(SLSQP ftol=1e-3)               : max %.3e  (bar: <= 1e-3)   f - f(SLSQP 1e-12): max %.3e min %.3e"
                  % ((cmds["cost"] - g[k + "f_loose"][mask]).max(), (cmds["cost"] - g[k + "f_tight"][mask]).max(),
                     (cmds["cost"] - g[k + "f_tight"][mask]).min()))
            print("P4  |u0 - u0(SLSQP ftol=1e-3)|_inf       : %s" % pct(du_l))
            print("    ... on the command                   : %s" % pct(dc_l))
            print("    reference vs itself, |u0(SLSQP 1e-3) - u0(SLSQP 1e-12)|_inf : %s" % pct(ref_ll))
            print("    ... on the command                   : %s" % pct(dc_ref))
            print("    iterations: mean %.1f max %d; converged %d/%d (status 1 = iteration cap: %d); SLSQP@1e-3: nit mean %.1f"
                  % (cmds["iterations"].mean(), cmds["iterations"].max(), (cmds["status"] == 0).sum(), len(cmds),
                     (cmds["status"] == 1).sum(), g[k + "nit_loose"][mask].mean()))


def other_parameter_sets():
    """G8: the reference's SLSQP solves at parameter sets away from the README's (general kernels)."""
    print("\n# other parameter sets (G8): 'cut' = vx/vy box cutting the speed disc; 'turn' = max_vel_theta 3, horizon 1.2 s, "
          "w_orient 2, w_costmap 0.3, w_control 0.1")
    g = util.load("g8_solves_params.npz")
    for pset in ("cut", "turn", "readme"):   # ("readme": the README's parameters at control_steps 16)
        for n in ((16,) if pset == "readme" else (3, 8)):
            k = "%s_n%d_" % (pset, n)
            params = util.params_from(g["param_keys"], g[k + "params"])
            probs = util.problems_from(g[k + "problems"])
            hm = g[k + "has_map"].astype(bool)
            for mask, cells, name in ((~hm, np.zeros_like(g[k + "cells"]), "zero costmap"), (hm, g[k + "cells"], "costmap")):
                pr = probs[mask]
                st, warm = synthetic.make_states(pr, n)
                with BatchSolver(params) as s:
                    s.set_costmap(cells, *g[k + "map_meta"])
                    cmds, x = s.solve(pr, st, warm)
                worse = cmds["cost"] - g[k + "f_loose"][mask]
                du = np.abs(x[:, :3] - g[k + "x_tight"][mask][:, :3]).max(axis=1)
                print("%-4s control_steps %d, %-12s: P3 f - f(SLSQP 1e-3): max %.3e, above 1e-3: %d of %d, below -1e-3: %d, median %.2e"
                      " | P2 |u0 - u0(SLSQP 1e-12)|: %s | iterations %.1f"
                      % (pset, n, name, worse.max(), (worse > 1e-3).sum(), len(worse), (worse < -1e-3).sum(), np.median(worse),
                         pct(du), cmds["iterations"].mean()))


def warm_starts():
    print("\n# warm starts: every call of the reference's recorded episodes (G4) solved from the reference's own state")
    for fixture in util.EPISODE_FIXTURES:
        g = util.load(fixture)
        params = util.params_from(g["param_keys"], g["params"])
        n = params["control_steps"]
        cmap = (g["cells"],) + tuple(g["map_meta"])
        probs = util.problems_from(g["problems"])
        n_ep, n_calls = probs.shape
        states, warm = abi.new_states(n_ep, n)
        df, dcmd, its = [], [], []
        with BatchSolver(params) as s:
            s.set_costmap(*cmap)
            for k in range(n_calls):
                fp = g["footprint"][:, k]
                rows = probs[:, k].copy()
                has = ~np.isnan(fp).any(axis=(1, 2))
                rows["footprint_cost"] = 0.0
                if has.any():
                    rows["footprint_cost"][has] = c_oracle.footprint_cost_batch(cmap, fp[has])
                cmds, x = s.solve(rows, states.copy(), warm.copy())
                df.append(cmds["cost"] - s.objective(rows, g["raw_x"][:, k]))
                moving = ~(g["collision"][:, k] | g["collision_footprint"][:, k])
                dcmd.append(np.where(moving, np.abs(cmds["vel"] - g["out"][:, k]).max(axis=1), 0.0))
                its.append(cmds["iterations"])
                s.postprocess(rows, states, warm, g["raw_x"][:, k], g["success"][:, k])
        df, dcmd, its = np.array(df), np.array(dcmd), np.array(its)
        print("\n## %s, control_steps %d: %d calls" % (fixture, n, df.size))
        print("P3  f(build) - f(reference raw x.x)      : max %.3e  median %.3e  (bar: <= 1e-3)" % (df.max(), np.median(df)))
        print("P4  |command - reference command|_inf    : %s   (calls not stopped by the collision latch)" % pct(dcmd.ravel()))
        print("    iterations: mean %.1f max %d" % (its.mean(), its.max()))


def flat_problem_drift():
    """How far the default stop rules leave the first control from a solve of the same kernel run to the end
    (window off, step tolerance 1e-9, cost tolerance 1e-12, 300 iterations) on 1024 zero-costmap problems --
    a broader sample than the reference fixtures can give (SLSQP to 1e-12 at control_steps 32 takes 20 s a
    solve).  The long-horizon problems have valleys: two neighbouring blocks can trade displacement at almost
    no cost, so u0 may sit 1e-2 away at an objective within 1e-4 of the minimum."""
    print("\n# default stop rules vs the same kernel run to the end, 1024 zero-costmap cold starts")
    for n in (3, 8, 32):
        cmap = synthetic.make_costmap(500, seed=0)
        zero = (np.zeros_like(cmap[0]),) + cmap[1:]
        pr = synthetic.make_problems(1024, 500, seed=77)
        res = []
        for over in ({}, dict(window_tolerance=-1.0, step_tolerance=1e-9, cost_tolerance=1e-12, max_iterations=300)):
            st, warm = synthetic.make_states(pr, n)
            with BatchSolver(dict(util.orc.make_params(control_steps=n), **over)) as s:
                s.set_costmap(*zero)
                res.append(s.solve(pr, st, warm))
        (c0, x0), (c1, x1) = res
        du = np.abs(x0[:, :3] - x1[:, :3]).max(axis=1)
        print("control_steps %2d: |u0 - u0(run to the end)|_inf : %s ; above 1e-3: %d of 1024; f - f(end): max %.2e; "
              "iterations %.1f vs %.1f" % (n, pct(du), (du > 1e-3).sum(), (c0["cost"] - c1["cost"]).max(),
                                         c0["iterations"].mean(), c1["iterations"].mean()))


def node_defaults():
    """G9: the node's own declared defaults (py:49-75, opt_tolerance 1e-5): P2 / P3 and the literal command gates."""
    print("\n# G9: the node's declared defaults (opt_tolerance 1e-5, every weight 0.5, w_footprint 2000, limits 0.5, horizon 0.5)")
    for fixture, prefix in util.G9_GROUPS:
        g, params, probs, hm = util.solve_group(fixture, prefix)
        n = params["control_steps"]
        for name, mask, cells in (("zero costmap", ~hm, np.zeros_like(g["cells"])), ("costmap", hm, g["cells"])):
            cmap = (cells,) + tuple(g["map_meta"])
            pr = probs[mask]
            st, warm = synthetic.make_states(pr, n)
            with BatchSolver(params) as s:
                s.set_costmap(*cmap)
                cmds, x = s.solve(pr, st, warm)
            ok = g["status_tight"][mask] == 0
            du = np.abs(x[:, :3] - g["x_tight"][mask][:, :3]).max(axis=1)
            worse = cmds["cost"] - g["f_loose"][mask]
            v_b, v_l, v_t = (command_of(params, cmap, pr, x), command_of(params, cmap, pr, g["x_loose"][mask]),
                             command_of(params, cmap, pr, g["x_tight"][mask]))
            d_bt, d_bl, d_lt = (np.abs(v_b - v_t).max(axis=1), np.abs(v_b - v_l).max(axis=1), np.abs(v_l - v_t).max(axis=1))
            print("\n## control_steps %d, %s, %d cases (SLSQP 1e-12 status 0 on %d); SLSQP as shipped (ftol 1e-5): %.1f iterations"
                  % (n, name, mask.sum(), ok.sum(), g["nit_loose"][mask].mean()))
            print("P2  |u0 - u0(SLSQP 1e-12)|_inf           : %s" % pct(du))
            print("P3  f - f(SLSQP as shipped, 1e-5)         : max %.3e  above 1e-3: %d   f - f(SLSQP 1e-12): max %.3e"
                  % (worse.max(), (worse > 1e-3).sum(), (cmds["cost"] - g["f_tight"][mask]).max()))
            print("L1  |command - command(SLSQP 1e-12)|      : %s" % pct(d_bt))
            print("L2  |command - command(SLSQP as shipped)| : %s ; within 1e-3: %d of %d" % (pct(d_bl), (d_bl <= 1e-3).sum(), len(d_bl)))
            print("    the reference as shipped vs its own converged answer: %s ; within 1e-3: %d of %d ; max(L2 - this) %.2e"
                  % (pct(d_lt), (d_lt <= 1e-3).sum(), len(d_lt), (d_bl - d_lt).max()))
            settled = d_lt <= 1e-4
            print("L3  where the reference as shipped is settled to 1e-4 (%d cases): max |command difference| %.2e; iterations %.1f"
                  % (settled.sum(), d_bl[settled].max() if settled.any() else float("nan"), cmds["iterations"].mean()))


def costmap_weight_sweep():
    print("\n# G8 mid: README parameters, w_costmap / w_trans = 0.10 ... 0.30, every case on the costmap (32 each)")
    for fixture, prefix in util.G8_MID_GROUPS:
        g, params, probs, hm = util.solve_group(fixture, prefix)
        cmap = (g["cells"],) + tuple(g["map_meta"])
        for method, tag in ((2, "dense Newton"), (3, "stage-wise")):
            if method == 2 and params["w_costmap"] > 0.25 * params["w_trans"]:
                print("ratio %.2f %-12s: not offered (NEO_MPC_ERR_UNSUPPORTED above w_costmap = w_trans / 4)" % (int(prefix[1:3]) / 100.0, tag))
                continue
            st, warm = synthetic.make_states(probs, 3)
            with BatchSolver(dict(params, method=method)) as s:
                s.set_costmap(*cmap)
                cmds, x = s.solve(probs, st, warm)
            worse = cmds["cost"] - g["f_loose"]
            print("ratio %.2f %-12s: P3 f - f(SLSQP 1e-3): max %.3e above 1e-3: %d of %d | f - f(SLSQP 1e-12): max %.2e median %.1e | iterations %.1f"
                  % (int(prefix[1:3]) / 100.0, tag, worse.max(), (worse > 1e-3).sum(), len(worse),
                     (cmds["cost"] - g["f_tight"]).max(), np.median(cmds["cost"] - g["f_tight"]), cmds["iterations"].mean()))


def long_horizon_unique_minimisers():
    g, params, probs, _ = util.solve_group("g3_solves_n32_zero.npz", "")
    zero = (np.zeros((200, 200), np.uint8),) + tuple(g["map_meta"])
    st, warm = synthetic.make_states(probs, 32)
    with BatchSolver(params) as s:
        s.set_costmap(*zero)
        cmds, x = s.solve(probs, st, warm)
    ok = g["status_tight"] == 0
    du = np.abs(x[:, :3] - g["x_tight"][:, :3]).max(axis=1)
    print("\n# G3 n32: 64 all-free-map problems at control_steps 32, SLSQP maxiter raised to 8000 (status 0 on %d; its iterations: "
          "median %d max %d)" % (ok.sum(), np.median(g["nit_tight"]), g["nit_tight"].max()))
    print("P2  |u0 - u0(SLSQP 1e-12)|_inf on the %d    : %s" % (ok.sum(), pct(du[ok])))
    print("P3  f - f(SLSQP 1e-3): max %.3e ; f - f(SLSQP 1e-12): max %.3e ; iterations %.1f max %d"
          % ((cmds["cost"] - g["f_loose"]).max(), (cmds["cost"] - g["f_tight"]).max(), cmds["iterations"].mean(), cmds["iterations"].max()))


def warm_drift():
    """Warm-started searches on an all-free map against the same kernel run to the end from the same state: 4096 robots
    after 12 closed-loop ticks (pose fixed, velocity = the previous command)."""
    print("\n# warm starts on an all-free map vs the same kernel run to the end (4096 robots after 12 ticks)")
    cmap = synthetic.make_costmap(500, seed=0)
    zero = (np.zeros_like(cmap[0]),) + cmap[1:]
    p = synthetic.make_problems(4096, 500, seed=77)
    st, warm = synthetic.make_states(p, 3)
    params = util.orc.make_params()
    with BatchSolver(params) as s:
        s.set_costmap(*zero)
        for _ in range(12):
            cm, _ = s.solve(p, st, warm)
            p["cur_vel"] = cm["vel"]
        c1, x1 = s.solve(p, st.copy(), warm.copy())
    with BatchSolver(dict(params, window_tolerance=-1.0, step_tolerance=1e-9, cost_tolerance=1e-12, max_iterations=400)) as s:
        s.set_costmap(*zero)
        c2, x2 = s.solve(p, st.copy(), warm.copy())
    du, dv, df = np.abs(x1[:, :3] - x2[:, :3]).max(axis=1), np.abs(c1["vel"] - c2["vel"]).max(axis=1), c1["cost"] - c2["cost"]
    print("|u0 - u0(run to the end)|: %s ; above 1e-3: %d of 4096" % (pct(du), (du > 1e-3).sum()))
    print("|command difference|     : %s ; above 1e-3: %d" % (pct(dv), (dv > 1e-3).sum()))
    print("f - f(run to the end)    : max %.2e median %.1e ; iterations %.1f vs %.1f" % (df.max(), np.median(df), c1["iterations"].mean(), c2["iterations"].mean()))


def held_out_sets():
    """G10: parameter sets nobody looked at while thresholds were tuned (round 4)."""
    print("\n# G10 held-out parameter sets (a, b: the round-3 judge's; c: a third), control_steps 3 / 5 / 8 / 12, 300 x 300 maps of other seeds")

    def solve(params, cmap, pr):
        st, warm = synthetic.make_states(pr, params["control_steps"])
        with BatchSolver(params) as s:
            s.set_costmap(*cmap)
            return s.solve(pr, st, warm)
    for fixture, groups, min_ok in (("g10_heldout.npz", util.G10_GROUPS, 8), ("g12_after_tuning.npz", util.G12_GROUPS, 8)):
        if fixture.startswith("g12"):
            print("\n# G12: three more parameter sets (d, e, f), control_steps 3 / 4 / 6 / 10 -- generated AFTER the last change "
                  "of round 4 to the search or a threshold")
        for name, n in groups:
            m = util.check_held_out_group(solve, name, n, p2_bar=1e-3, fixture=fixture, min_ok=min_ok)
            print("set %s control_steps %2d: P2 max|u0 - u0(SLSQP 1e-12)| %.2e over %d unique cases (%d not unique: %.2e) ; "
                  "P3 max f - f(SLSQP as shipped): all-free map %.2e costmap %.2e ; iterations %.1f / %.1f"
                  % (name, n, m["p2"], m["p2_cases"], m["not_unique"], m["p2_not_unique"], m["p3_free"], m["p3_map"], m["it_free"], m["it_map"]))


def warm_gate():
    """G11: the deployed (warm-started) mode against the reference run to convergence (round 4)."""
    print("\n# G11 warm-started ticks: K1 at the README tolerance from the reference's own state vs the reference's CONVERGED command "
          "(opt_tolerance 1e-12, maxiter 500, all-free map)")
    for fixture in util.G11_FIXTURES + util.G13_FIXTURES:
        if fixture == util.G13_FIXTURES[0]:
            print("# G13: the same protocol at G10's set \"a\" (heavy control weight, box cutting the disc), generated after the tuning stopped")
        solvers = {}

        def get(params, cmap):
            if "s" not in solvers:
                solvers["s"] = BatchSolver(params)
                solvers["s"].set_costmap(*cmap)
            return solvers["s"]
        dv, du, its, settled = util.warm_gate(lambda p, c, r, st, wm: get(p, c).solve(r, st, wm),
                                     lambda p, c, r, st, wm, x, ok: get(p, c).postprocess(r, st, wm, x, ok), fixture)
        solvers["s"].close()
        print("%s: %d ticks the reference converged on" % (fixture, dv.size))
        print("|command - reference command|_inf : %s ; above 1e-3: %d (%.3f %%) -- %d of them on the %d ticks the fixture flags "
              "not settled (the reference's answer, the same solve taken up again from it and one from zeros disagree by > 1e-4)"
              % (pct(dv), (dv > 1e-3).sum(), 100.0 * (dv > 1e-3).mean(), (dv[~settled] > 1e-3).sum(), (~settled).sum()))
        print("|u0 - reference u0|_inf           : %s ; above 1e-3: %d" % (pct(du), (du > 1e-3).sum()))
        print("iterations: mean %.2f max %d" % (its.mean(), its.max()))


def random_sets():
    """G14 / G15: miss counts over 48 + 64 random parameter sets."""

    def solve(params, cmap, pr):
        st, warm = synthetic.make_states(pr, params["control_steps"])
        with BatchSolver(params) as s:
            s.set_costmap(*cmap)
            return s.solve(pr, st, warm)
    for fixture in sorted(util.RANDOM_SETS):
        print("\n# %s: %d RANDOM parameter sets x 24 cold problems against the reference (tests/util.random_sets_miss_rates)"
              % (fixture, util.RANDOM_SETS[fixture]))
        m = util.random_sets_miss_rates(solve, fixture)
        print("P3 misses (objective more than 1e-3 above SLSQP as shipped): %d of %d all-free-map cases, %d of %d costmap cases; SLSQP as "
              "shipped more than 1e-3 above the build: %d of %d" % (m["p3_miss_free"], m["cases_free"], m["p3_miss_map"], m["cases_map"],
                                                                   m["ref_worse"], m["cases_free"] + m["cases_map"]))
        print("P2 misses (first control more than 1e-3 from SLSQP run to the end): %d of %d unique cases (worst %.2e); %d more status-0 "
              "cases are not unique by the reference's own answers (worst distance there %.2e); worst P3 margin %.2e; %s"
              % (m["p2_miss"], m["p2_cases"], m["p2_worst"], m["not_unique"], m["p2_not_unique_worst"], m["p3_worst"], m["misses"]))


if __name__ == "__main__":
    cold_starts()
    long_horizon_unique_minimisers()
    other_parameter_sets()
    costmap_weight_sweep()
    node_defaults()
    warm_starts()
    flat_problem_drift()
    warm_drift()
    held_out_sets()
    warm_gate()
    random_sets()
