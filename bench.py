#!/usr/bin/env python3
"""bench.py -- MPC solves/sec of the batched HIP solver (BASELINE.json metric).

A "step" is one pass of the hot path (K1: full `optimizer()` equivalent -- solve +
low-pass + collision check + acceleration clamp + warm-start shift) over one batch of
synthetic instances that is already resident in HBM.  Workload = BASELINE config 2:
4 096 independent instances per GPU, control_steps=3, 500x500 costmap, README params.

    python bench.py [--gpus N] [--steps K] [--warmup W]          (N > 1 without WORLD_SIZE: spawns the N ranks itself)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Multi-GPU: weak scaling, instances shard embarrassingly (one process per GPU, each
with its own 4 096 instances and a replica of the costmap); the only exchange is one
RCCL all-gather of the (vx, vy, omega) commands per step.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# BASELINE.md "Algorithmic bytes per solve": inputs (17+3N)*4 + outputs (3+3N+1)*4 + reach tile 729
ALGO_BYTES = {3: 885, 8: 1005, 32: 1581}
SIMD_CLOCK_GHZ = 2.4   # MI355X peak engine clock (MI355X_MICROARCH.md)
HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def readme_params(control_steps):
    from neo_mpc_planner2_amd.mpc_optimization_server import README_PARAMS
    p = dict(README_PARAMS)
    p["control_steps"] = control_steps
    return p


def cpu_baseline(params, cmap, probs, seconds=15.0):
    """The reference's path restated (oracle/mpc_oracle.py): SciPy SLSQP at ftol=opt_tolerance on
    the Python objective, cold start, on a bounded sample of the SAME workload, 1 core."""
    from oracle import mpc_oracle as orc
    cm = orc.Costmap(*cmap)
    n = params["control_steps"]
    done, t0 = 0, time.perf_counter()
    while done < len(probs):
        row = probs[done]
        prob = orc.Problem(row["cur_xy"], row["cur_q"], row["carrot_xy"], row["carrot_q"], row["goal_xyz"],
                           row["goal_q"], row["cur_vel"], float(row["control_interval"]), float(row["delta_t"]))
        state = orc.ServerState(n)
        state.old_goal = prob.goal_key()
        state.last_control = list(prob.cur_vel)
        state.waiting_time = 0.0
        orc.optimizer_step(state, prob, params, cm)
        done += 1
        if time.perf_counter() - t0 > seconds:
            break
    dt = time.perf_counter() - t0
    return done / dt, done, dt


def pcie_inclusive(solver, probs, st, warm, n, seconds=3.0):
    """SURVEY 8d's metric with the transfers inside: the host-buffer entry point neo_mpc_solve_batch (H2D of the
    requests / state / warm start, K1, D2H of commands / state / warm start / solution per call) on the SAME
    instances, timed on the host clock around the synchronous call.  Reported beside `value`, never as it."""
    count = len(probs)
    reps, t_all = 0, 0.0
    solver.solve(probs, st.copy(), warm.copy())            # staging buffers allocated
    while t_all < seconds and reps < 200:
        s_i, w_i = st.copy(), warm.copy()
        t0 = time.perf_counter()
        solver.solve(probs, s_i, w_i)
        t_all += time.perf_counter() - t0
        reps += 1
    res = {"value": count * reps / t_all, "unit": "solves/s", "ms_per_call": 1e3 * t_all / reps, "calls": reps,
           # (round 4: the device reads 216 of a request's 256 bytes and 104 of a state record's 128; it writes back 48 bytes
           # of state per tick -- old_goal only when it changed -- + warm start + command + solution; the staged path still
           # copies whole records)
           "bytes_in_per_call": count * (256 + 128 + 24 * n), "bytes_out_per_call": count * (48 + 128 + 48 * n),
           "bytes_in_per_call_in_place": count * (216 + 104 + 24 * n), "bytes_out_per_call_in_place": count * (48 + 48 + 48 * n),
           "what": "neo_mpc_solve_batch on host buffers (pageable NumPy arrays), same instances, cold start"}
    # the same call on page-locked host buffers (what a fleet server that owns its request arena would hand over):
    # every transfer is then a DMA queued behind / in front of the kernel, one wait per call
    try:
        import torch
        from neo_mpc_planner2_amd import abi

        def pinned(a):
            t = torch.empty(a.nbytes, dtype=torch.uint8).pin_memory()
            v = t.numpy().view(a.dtype).reshape(a.shape)
            v[...] = a
            return v
        p_probs, p_st, p_warm = pinned(np.ascontiguousarray(probs)), pinned(st), pinned(warm)
        p_cmd = pinned(np.zeros(count, dtype=abi.COMMAND_DTYPE))
        p_sol = pinned(np.zeros((count, 3 * n)))
        ref_cmd = None
        what = {"zerocopy": "page-locked host arrays (torch pin_memory), K1 reads the records and writes the results "
                            "in place over PCIe: no copy, one launch, one wait",
                "zerocopy_out": "page-locked arrays: inputs up as three DMA copies, results written in place by K1",
                "staged": "page-locked arrays staged through device memory: three DMA copies in, K1, four out"}
        for mode in ("zerocopy", "zerocopy_out", "staged"):
            solver.set_host_path(mode)
            p_st[...] = st
            p_warm[...] = warm
            solver.solve(p_probs, p_st, p_warm, out=(p_cmd, p_sol))
            if ref_cmd is None:
                ref_cmd = p_cmd["vel"].copy()
            same = bool((p_cmd["vel"] == ref_cmd).all())     # (one kernel, three ways of feeding it)
            # three rounds, the median round reported (a synchronous host call is exposed to whatever else the host does:
            # one run of r03 had a single round at 0.36 ms per call between rounds at 0.155)
            rounds = []
            for _ in range(3):
                reps, t_all = 0, 0.0
                while t_all < seconds / 6 and reps < 100:
                    p_st[...] = st
                    p_warm[...] = warm
                    t0 = time.perf_counter()
                    solver.solve(p_probs, p_st, p_warm, out=(p_cmd, p_sol))
                    t_all += time.perf_counter() - t0
                    reps += 1
                rounds.append((t_all / reps, reps))
            per_call, reps = sorted(rounds)[1]
            rec = {"value": count / per_call, "unit": "solves/s", "ms_per_call": 1e3 * per_call, "calls": reps,
                   "rounds_ms_per_call": [1e3 * r[0] for r in rounds],
                   "commands_identical": same, "what": what[mode]}
            if mode == "zerocopy":
                res["pinned"] = rec          # the default path of page-locked batches
            else:
                res["pinned_" + mode] = rec
        solver.set_host_path("auto")
        # Two batches in flight (neo_mpc_solve_batch_begin / _wait = the reference's async_send_request / result.get(),
        # cpp:248-250): the same instances as two page-locked copies, A and B; while one is on the GPU the host resets and
        # resubmits the other.  Host clock around the whole loop, the resets of the cold-start state inside it.
        q_probs, q_st, q_warm = pinned(np.ascontiguousarray(probs)), pinned(st), pinned(warm)
        q_cmd, q_sol = pinned(np.zeros(count, dtype=abi.COMMAND_DTYPE)), pinned(np.zeros((count, 3 * n)))
        sets = [(p_probs, p_st, p_warm, p_cmd, p_sol), (q_probs, q_st, q_warm, q_cmd, q_sol)]
        tickets = [None, None]
        for k, a in enumerate(sets):
            a[1][...] = st
            a[2][...] = warm
            tickets[k] = solver.solve_begin(a[0], a[1], a[2], out=(a[3], a[4]))
        rounds, total = [], 0
        for _ in range(3):   # (three rounds, the median one reported, as above)
            calls, t0 = 0, time.perf_counter()
            while time.perf_counter() - t0 < seconds / 6 and calls < 200:
                k = (total + calls) & 1
                solver.solve_wait(tickets[k])
                a = sets[k]
                a[1][...] = st
                a[2][...] = warm
                tickets[k] = solver.solve_begin(a[0], a[1], a[2], out=(a[3], a[4]))
                calls += 1
            rounds.append(((time.perf_counter() - t0) / calls, calls))
            total += calls
        for t in tickets:
            solver.solve_wait(t)
        per_call, calls = sorted(rounds)[1]
        res["pinned_two_in_flight"] = {
            "value": count / per_call, "unit": "solves/s", "ms_per_call": 1e3 * per_call, "calls": calls,
            "rounds_ms_per_call": [1e3 * r[0] for r in rounds],
            "commands_identical": bool((p_cmd["vel"] == ref_cmd).all() and (q_cmd["vel"] == ref_cmd).all()),
            "what": "neo_mpc_solve_batch_begin / _wait, two page-locked batches of the same instances alternating: one on the "
                    "GPU (worked on in place) while the host resets and resubmits the other; resets inside the timed loop"}
    except Exception as e:   # (the pageable figure stands on its own)
        res["pinned"] = {"error": str(e)}
    return res


def cpu_baseline_all_cores(workload, seconds=8.0, cap=64):
    """SURVEY 8d B1's "per-core processes": the same SciPy port in `procs` fresh interpreters (no fork: the HIP runtime is
    live in this one), each on its own slice of the workload for `seconds`; the aggregate of their own rates."""
    import subprocess
    procs = max(1, min(usable_cpus(), cap))
    cmd = [sys.executable, os.path.abspath(__file__), "--workload", workload, "--cpu-worker"]
    env = dict(os.environ, OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1", MKL_NUM_THREADS="1")
    kids = [subprocess.Popen(cmd + ["%d:%d:%g" % (k, procs, seconds)], stdout=subprocess.PIPE, env=env) for k in range(procs)]
    rate, done = 0.0, 0
    for kid in kids:
        line = kid.communicate(timeout=seconds * 6 + 120)[0].decode().strip().splitlines()[-1]
        rec = json.loads(line)
        rate += rec["rate"]
        done += rec["done"]
    return rate, done, procs


def usable_cpus():
    """CPUs this process may actually use: affinity mask, cut by the cgroup's CPU quota (a container on a 256-thread
    host may own eight of them)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(float(txt[0]) / float(txt[1]) + 0.5)))
            else:
                quota = float(txt[0])
                period = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if quota > 0:
                    n = min(n, max(1, int(quota / period + 0.5)))
            break
        except (OSError, ValueError, IndexError):
            continue
    return n


def cpu_worker(spec, workload):
    """One process of cpu_baseline_all_cores: slice k of `procs` of the workload, no GPU, no torch."""
    from neo_mpc_planner2_amd import synthetic
    k, procs, seconds = spec.split(":")
    k, procs, seconds = int(k), int(procs), float(seconds)
    cfg = dict(synthetic.CONFIGS["C2" if workload == "C4" else workload])
    cmap = synthetic.make_costmap(cfg["map_size"], seed=0)
    probs = synthetic.make_problems(min(cfg["batch"], 4096), cfg["map_size"], seed=1000)
    mine = probs[k::procs]
    cpu_baseline(readme_params(cfg["control_steps"]), cmap, mine[:2], seconds)            # (imports, first-call set-up)
    mine = np.concatenate([mine] * max(1, int(2000 * seconds / max(1, len(mine)))))       # enough to fill the time
    rate, done, secs = cpu_baseline(readme_params(cfg["control_steps"]), cmap, mine, seconds)
    print(json.dumps({"rate": rate, "done": done, "seconds": secs}))


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_mirror_rate(params, cmap, probs, st, warm):
    """Secondary: the build's own algorithm on the host cores (oracle/mpc_oracle.c, OpenMP)."""
    from oracle import c_oracle
    c_oracle.load()
    c_oracle.set_threads(usable_cpus())     # (the OpenMP default is every hardware thread of the host)
    t0 = time.perf_counter()
    c_oracle.solve_batch(params, cmap, probs, st.copy(), warm.copy())
    return len(probs) / (time.perf_counter() - t0)


def source_sha():
    """sha256 over the device sources K1 is built from: a PMC figure in profiles/hbm_traffic.json is only
    reported while the kernel it was measured on is the kernel that runs."""
    import hashlib
    d = os.path.join(ROOT, "neo_mpc_planner2_amd", "csrc")
    h = hashlib.sha256()
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".h", ".cpp")) or name == "Makefile":   # (the Makefile carries per-file flags)
            h.update(name.encode())
            h.update(open(os.path.join(d, name), "rb").read())
    return h.hexdigest()[:16]



def pmc_entry(workload, batch):
    """The committed PMC passes of a workload (profiles/hbm_traffic.json): reported only while the device sources are the
    ones the counters were collected on (source_sha) and the batch is the config's; (None, reason) otherwise."""
    tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    if not os.path.exists(tpath):
        return None, "no profiles/hbm_traffic.json"
    try:
        entry = json.load(open(tpath)).get(workload)
    except Exception as e:
        return None, "profiles/hbm_traffic.json unreadable: %s" % e
    if not isinstance(entry, dict):
        return None, "no PMC entry for this workload in profiles/hbm_traffic.json"
    if entry.get("source_sha") != source_sha():
        return None, "stale: PMC passes ran on source_sha %s, this build is %s" % (entry.get("source_sha"), source_sha())
    if entry.get("batch") != batch:
        return None, "PMC passes ran at batch %s" % entry.get("batch")
    return entry, entry.get("note")


#: issue cycles per wave64 instruction by counter class (tools/mb_valu_rates.hip, mb_select_rates.hip; table in
#: profiles/r03_b_ab_experiments.txt): float64 and transcendental / conversion instructions 4, plain float32 fma / add / mul
#: and 32-bit integer arithmetic 2.  What no typed counter covers -- moves, selects, compares, lane reads, DPP -- was
#: measured at 2 (moves) to 4 (compares, SGPR-masked selects, v_readlane, DPP): priced at both, `frac_low` / `frac_high`.
VALU_CYCLES = {"FMA_F64": 4, "ADD_F64": 4, "MUL_F64": 4, "TRANS_F64": 4, "TRANS_F32": 4, "CVT": 4, "INT64": 4,
               "FMA_F32": 2, "ADD_F32": 2, "MUL_F32": 2, "INT32": 2}


def valu_issue(entry, k_ms):
    """what actually bounds K1: VALU issue.  SQ_INSTS_VALU of the committed PMC pass priced in issue
    cycles by instruction class (VALU_CYCLES; the class counts come from the type-mix PMC passes) over 1024 SIMDs x this
    run's kernel time at the clock MEASURED in the PMC pass (GRBM_GUI_ACTIVE per XCD over the dispatch's own duration),
    not the 2.4 GHz name-plate.  `frac_at_4_cycles` prices every instruction at 4 cycles at the name-plate clock (the
    figure of rounds 1-2)."""
    if not entry or not entry.get("valu_insts"):
        return None
    total, mix = entry["valu_insts"], entry.get("valu_mix") or {}
    clock = entry.get("clock_ghz_measured") or SIMD_CLOCK_GHZ
    simd_cycles = 1024 * k_ms * 1e-3 * clock * 1e9
    typed = {k: mix.get("SQ_INSTS_VALU_" + k, 0.0) for k in VALU_CYCLES}
    typed_cycles = sum(VALU_CYCLES[k] * v for k, v in typed.items())
    untyped = max(0.0, total - sum(typed.values()))
    out = {"insts_per_launch": total, "typed_insts_per_launch": sum(typed.values()), "untyped_insts_per_launch": untyped,
           "simds": 1024, "clock_ghz": clock, "clock_ghz_measured": entry.get("clock_ghz_measured"),
           "clock_note": entry.get("clock_note"),
           "frac_low": (typed_cycles + 2 * untyped) / simd_cycles, "frac_high": (typed_cycles + 4 * untyped) / simd_cycles,
           "frac_at_4_cycles": total * 4 / (1024 * k_ms * 1e-3 * SIMD_CLOCK_GHZ * 1e9)}
    out["frac"] = 0.5 * (out["frac_low"] + out["frac_high"])
    return out


def roofline(workload, batch, n, k_ms, launches_timed=None, params_over=None):
    """The roofline object of K1, named after the resource that binds it: VALU issue -- `achieved` = the
    launch's vector instructions priced in issue cycles (valu_issue(): the mean of the low and high pricing) over the kernel's
    mean duration, `peak` = 1024 SIMDs x the clock measured in the PMC pass.  The HBM object the north star asks for rides
    inside as `hbm`: `achieved` = algorithmic bytes per launch over the kernel's mean duration; `traffic` = HBM bytes per
    launch from the committed PMC passes, CORRECTED with the calibration of the counters on K1's own access shapes
    (tools/mb_k1_traffic.hip; raw FETCH_SIZE / WRITE_SIZE beside it).  Without a PMC pass of this build (stale sources, other
    parameter set) there is no instruction count to price: `bound` falls back to "hbm" and the HBM figures are the object."""
    # (control_steps 3 under AUTO: the routed kernel -- direction by neighbourhood; a pinned method: the one-direction kernel)
    kernel_name = "k_solve_routed" if n == 3 and not (params_over or {}).get("method") else "k_solve"
    algo = ALGO_BYTES.get(n, (17 + 3 * n) * 4 + (3 + 3 * n + 1) * 4 + 729)
    achieved = algo * batch / (k_ms * 1e-3) / 1e9
    entry, note = pmc_entry(workload, batch) if not params_over else (None, "other parameter set")
    h = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
         "frac": achieved / HBM_PEAK_GBS, "traffic": None, "traffic_note": note, "algorithmic_bytes_per_solve": algo}
    if entry:
        h["traffic"] = entry.get("hbm_bytes_calibrated") or entry.get("hbm_bytes")
        h["traffic_raw"] = entry.get("hbm_bytes")
        h["traffic_note"] = ("calibrated: " + entry["calibration_note"]) if entry.get("hbm_bytes_calibrated") else \
            ("raw counters (no calibration entry): " + str(entry.get("note")))
        h["traffic_over_algorithmic"] = h["traffic"] / (algo * batch) if h["traffic"] else None
    v = valu_issue(entry, k_ms)
    if v:
        peak = v["simds"] * v["clock_ghz"]
        r = {"bound": "valu_issue", "achieved": v["frac"] * peak, "peak": peak, "unit": "G issue-cycles/s", "frac": v["frac"],
             "frac_low": v["frac_low"], "frac_high": v["frac_high"],
             "what": "vector instructions per launch priced in issue cycles by class (bench.VALU_CYCLES; untyped ones at 2 and "
                     "at 4: frac_low / frac_high) over the kernel's duration, against 1024 SIMDs x the measured clock"}
    else:
        r = {k: h[k] for k in ("bound", "achieved", "peak", "unit", "frac")}
        r["what"] = "no PMC pass of this build and workload to price the instruction stream with (%s): HBM figures" % note
    r.update({"traffic": h["traffic"], "traffic_note": h["traffic_note"], "kernel": kernel_name, "kernel_ms": k_ms, "hbm": h})
    if launches_timed is not None:
        r["kernel_ms_launches_timed"] = launches_timed
    return r, entry


def other_workload(name, dev, local_rank, steps=3, warmup=1, params_over=None, label=None):
    """One more BASELINE config on this GPU, HBM-resident, cold start, `steps` launches timed with dispatch-stamped
    events: so that the driver's line carries C3 / C5 / the C4 shard (and the parameter sets that take the general
    kernels) next to the headline."""
    import torch
    from neo_mpc_planner2_amd import synthetic
    from neo_mpc_planner2_amd.solver import BatchSolver, DeviceBatch
    cfg = dict(synthetic.CONFIGS["C2" if name == "C4" else name])
    if name == "C4":
        cfg["batch"] = synthetic.CONFIGS["C4"]["batch"] // 8
    n = cfg["control_steps"]
    params = readme_params(n)
    params.update(params_over or {})
    cmap = synthetic.make_costmap(cfg["map_size"], seed=0)
    probs = synthetic.make_problems(cfg["batch"], cfg["map_size"], seed=1000)
    st, warm = synthetic.make_states(probs, n)
    with BatchSolver(params, device=local_rank) as solver:
        solver.set_costmap(torch.from_numpy(cmap[0]).to(dev), *cmap[1:])
        base = DeviceBatch(probs, st, warm, dev, want_solution=False)
        sets = [base.fresh_state() for _ in range(steps + warmup)]
        stream = torch.cuda.current_stream()
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in sets]
        for e0, e1 in evs:
            e0.record(stream)
            e1.record(stream)
        torch.cuda.synchronize()
        t0 = None
        stamped = []
        for i, b in enumerate(sets):
            if i == warmup:
                torch.cuda.synchronize()
                t0 = time.perf_counter()
            # (4096-instance launches: the event pair on the first timed launch only -- a stamped launch holds the stream
            # ~7 us longer than a plain one, 3-6 % of these kernels)
            timed = i >= warmup and (cfg["batch"] > 8192 or i == warmup)
            solver.solve_device(base.problems, b.states, b.warm, b.commands, velocities=b.vel,
                                events=evs[i] if timed else None)
            stamped.append(timed)
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        k_ms = float(np.mean([a.elapsed_time(b) for (a, b), on in zip(evs, stamped) if on]))
        cmds = sets[-1].commands_host()
    roof, entry = roofline(name, cfg["batch"], n, k_ms, params_over=params_over)
    return {"workload": label or name, "batch": cfg["batch"], "control_steps": n, "map_size": cfg["map_size"],
            "steps": steps, "value": cfg["batch"] * steps / elapsed, "unit": "solves/s", "ms_per_step": 1e3 * elapsed / steps,
            "kernel_ms": k_ms,
            "solver": {"mean_iterations": float(cmds["iterations"].mean()), "max_iterations_seen": int(cmds["iterations"].max()),
                       "converged_frac": float((cmds["status"] == 0).mean()), "status_max_iter": int((cmds["status"] == 1).sum())},
            "roofline": roof, "valu_issue": valu_issue(entry, k_ms)}


#: parameter sets away from the README's that take the GENERAL (non-"tame") kernels -- the ones the G8 fixtures pin:
#: the vx/vy box cutting the speed disc; a fast-turning robot with a heavy costmap weight over a 1.2 s horizon
GENERAL_SETS = {
    "C2/cut": dict(max_vel_trans=0.7, max_vel_x=0.4, min_vel_x=-0.2, max_vel_y=0.65, min_vel_y=-0.65),
    "C2/turn": dict(max_vel_theta=3.0, min_vel_theta=-3.0, w_orient=2.0, w_costmap=0.3, w_control=0.1,
                    prediction_horizon=1.2),
}


def warm_tick(dev, local_rank, ticks=60):
    """The deployed mode: the C2 fleet (4096 robots) in a closed 30 Hz control loop, state resident, every tick
    warm-started the reference's way (py:397-400) -- ms per tick and iterations over the warm ticks."""
    import torch
    from neo_mpc_planner2_amd import fleet, synthetic
    from neo_mpc_planner2_amd.solver import BatchSolver, DeviceBatch
    cfg, cmap, probs, st, warm = synthetic.make_workload("C2", seed=0)
    with BatchSolver(readme_params(3), device=local_rank) as solver:
        solver.set_costmap(torch.from_numpy(cmap[0]).to(dev), *cmap[1:])
        b = DeviceBatch(probs, st, warm, dev, want_solution=False)
        plain = fleet.summary(fleet.closed_loop(solver, b, ticks))
        # ... and as a fleet server runs it: every 5th tick the dispatch order of the following ticks is rebuilt from that
        # tick's iteration counts (neo_mpc_balance_dispatch_device; results bit for bit the same -- tests/test_gpu_edges.py)
        b = DeviceBatch(probs, st, warm, dev, want_solution=False)
        res = fleet.summary(fleet.closed_loop(solver, b, ticks, balance_every=5))
    res["what"] = ("C2 fleet in closed loop: 4096 robots, 30 Hz, robots moved by their own commands, warm start = the "
                   "previous solution shifted by one control step; kernel ms between events around each tick's launch; "
                   "dispatch order rebuilt every 5th tick from the iteration counts (their exponential average over the rebuilds; balanced dispatch), "
                   "`launch_order`: the same loop without")
    res["solves_per_s"] = 4096 / (1e-3 * res["ms_per_tick_median_incl_order"])
    res["launch_order"] = {k: plain[k] for k in ("ms_per_tick_median", "ms_per_tick_max", "mean_iterations", "max_iterations_median",
                                                 "max_iterations_max")}
    return res


def spawn_ranks(n):
    """`python bench.py --gpus N` with no launcher around it: re-exec under torch.distributed.run, one rank per
    GPU of this node (RCCL over xGMI).  Fails loudly when the node has fewer devices."""
    import socket
    import subprocess
    import torch
    have = torch.cuda.device_count()
    if have < n and os.environ.get("NEO_MPC_BENCH_SHARE_DEVICE") != "1":
        raise SystemExit("bench.py --gpus %d: only %d HIP device(s) visible" % (n, have))
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    raise SystemExit(subprocess.call(cmd, env=env))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="C2", choices=["C2", "C3", "C4", "C5"],
                    help="BASELINE config: C2 (default, the metric's), C3, C5; C4 = its per-GPU shard "
                         "(262 144 instances of the C2 problem per GPU; with --gpus 8 the full 2 097 152)")
    ap.add_argument("--batch", type=int, default=None, help="instances per GPU (default: the config's)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pcie", action="store_true", help="skip the host-buffer (PCIe-inclusive) leg")
    ap.add_argument("--no-others", action="store_true",
                    help="skip the other BASELINE configs (C3, C5, C4 shard, general-kernel parameter sets) and the "
                         "closed-loop warm ticks that ride in the default line")
    ap.add_argument("--cpu-worker", default=None, help=argparse.SUPPRESS)   # internal: one process of the all-cores CPU leg
    ap.add_argument("--max-iterations", type=int, default=None, help="study knob: cap the solver iterations")
    ap.add_argument("--control-steps", type=int, default=None, help="study knob: override the config's control_steps")
    ap.add_argument("--stamp-every", type=int, default=0,
                    help="single GPU: every how-manyth launch of the timed region carries the HIP event pair that times "
                         "the kernel (a stamped launch costs ~7 us of event handling on the stream; with --gpus > 1 every "
                         "launch is stamped, the all-gather waits on the stop event); 0 = every 8th, more often when "
                         "that would leave fewer than five timed launches")
    ap.add_argument("--settle-ms", type=float, default=40.0,
                    help="untimed launches for this many milliseconds ahead of the warm-up steps (clock ramp after idle; 0 = off)")
    ap.add_argument("--streams", type=int, default=1,
                    help="study knob (NOT the headline): consecutive steps alternate over this many streams, so "
                         "the next batch starts while the previous one's stragglers finish (independent fleets)")
    ap.add_argument("--method", type=int, default=None, help="study knob: 1 = L-BFGS, 2 = Newton (control_steps <= 8)")
    args = ap.parse_args()
    if args.cpu_worker:
        return cpu_worker(args.cpu_worker, args.workload)

    import torch
    import torch.distributed as dist
    from neo_mpc_planner2_amd import synthetic
    from neo_mpc_planner2_amd.solver import BatchSolver, DeviceBatch
    from neo_mpc_planner2_amd.sharding import gather_commands

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn_ranks(args.gpus)      # does not return
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    # NEO_MPC_BENCH_SHARE_DEVICE=1: every rank on device 0 -- the whole world > 1 path (spawned ranks, per-rank seeds, the
    # all-gather on device tensors, the MAX reduce, rccl.per_rank, the JSON line) on a box with ONE GPU.  RCCL refuses two
    # ranks on one device, so the collective goes through gloo (host-staged in sharding.gather_commands); the numbers of
    # such a run measure nothing, its `rccl.gather_check` and the shape of the line are what it is for.
    share = os.environ.get("NEO_MPC_BENCH_SHARE_DEVICE") == "1" and world > 1
    if share:
        local_rank = 0
    if torch.cuda.device_count() <= local_rank:
        raise SystemExit("rank %d: HIP device %d is not visible (%d devices)" % (rank, local_rank, torch.cuda.device_count()))
    torch.cuda.set_device(local_rank)
    dev = "cuda:%d" % local_rank
    # NEO_MPC_BENCH_FORCE_DIST=1 exercises the RCCL path with a single rank (used to smoke-test the
    # multi-GPU code on a 1-GPU box); the JSON line then still reports n_gpus = 1
    use_dist = world > 1 or os.environ.get("NEO_MPC_BENCH_FORCE_DIST") == "1"
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if share:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device(dev))
    comm_dev = "cpu" if share else dev      # where the small bookkeeping collectives live (gloo gathers host tensors only)

    cfg = dict(synthetic.CONFIGS[args.workload])
    if args.workload == "C4":
        cfg["batch"] //= 8          # the per-GPU shard of BASELINE config 4 (2 097 152 over 8 GPUs)
    if args.batch:
        cfg["batch"] = args.batch
    if args.control_steps:
        cfg["control_steps"] = args.control_steps
    n = cfg["control_steps"]
    params = readme_params(n)
    if args.max_iterations:
        params["max_iterations"] = args.max_iterations
    if args.method is not None:
        params["method"] = args.method
    cmap = synthetic.make_costmap(cfg["map_size"], seed=0)                 # shared map, replicated
    probs = synthetic.make_problems(cfg["batch"], cfg["map_size"], seed=1000 + rank)
    st, warm = synthetic.make_states(probs, n)

    solver = BatchSolver(params, device=local_rank)
    solver.set_costmap(torch.from_numpy(cmap[0]).to(dev), *cmap[1:])
    # one fresh (state, warm start, command) set per step, allocated up front (about 1 MB per
    # set at C2), so every step solves the same cold problems and nothing but the hot path runs
    # inside the timed region; the problems themselves are read-only and shared
    base = DeviceBatch(probs, st, warm, dev, want_solution=False)
    sets = [base.fresh_state() for _ in range(args.steps)]
    warm_sets = [base.fresh_state() for _ in range(max(1, min(args.warmup, 8)))]
    # RCCL all-gather of the commands: a ring of output buffers so that tick i's gather (on RCCL's
    # stream) overlaps tick i+1's solve kernel
    ring = 4
    gathered = [torch.empty((world, cfg["batch"], 3), dtype=torch.float64, device=dev) for _ in range(ring)] \
        if use_dist else None
    # a stream of its own for the hot path: on the legacy default stream the next tick's K1 ends up ordered
    # behind the previous tick's gather (measured with rocprofv3 --kernel-trace: 35 us between launches)
    torch.cuda.synchronize()
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    comm_stream = torch.cuda.Stream(device=dev) if use_dist else None
    pending = []

    gather_evs = []

    def exchange(i, b, done, timed=False):
        # The collective is issued from a side stream that waits for `done` (the event recorded after
        # this tick's K1), so the solve stream itself carries nothing but K1 and its two timing events:
        # every extra event record between two launches costs ~4 us of barrier-packet latency.  No
        # per-tick wait on the solve stream either: gathers are ordered among themselves on RCCL's
        # stream, so a ring buffer is rewritten only after its previous gather; handles are waited
        # for once, before the timed region closes.
        comm_stream.wait_event(done)
        with torch.cuda.stream(comm_stream):
            if timed:   # (events on the side stream: they put nothing between two K1 launches)
                g0, g1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                g0.record(comm_stream)
            _, work = gather_commands(b.vel, gathered[i % ring], async_op=True)   # packed by K1, no copy
            if timed:
                work.wait()
                g1.record(comm_stream)
                gather_evs.append((g0, g1))
        pending.append(work)

    def drain():
        while pending:
            pending.pop(0).wait()

    if use_dist:   # communicator set-up happens here, never inside the timed region
        ev = torch.cuda.Event()
        ev.record(stream)
        exchange(0, warm_sets[0] if warm_sets else sets[0], ev)
        drain()
    # HIP events per launch, stamped by the dispatch itself (hipExtLaunchKernel through
    # neo_mpc_solve_batch_device_timed): separate event records would put two barrier packets between
    # consecutive K1 launches.  (Recorded once here, ahead of the warm-up, so that the handles exist and
    # nothing but the synchronisation lies between the warm-up steps and the timed ones.)
    # Single GPU: every `stamp`-th launch carries a pair -- a stamped launch costs the stream ~7 us more than a plain one
    # (measured: 0.096-0.100 against 0.089-0.091 ms per step with all / none of 40 launches stamped), and the timed region
    # is the job, not its instrumentation; the kernel's duration is the mean over the stamped launches.
    stamp = 1 if use_dist else max(1, min(args.stamp_every or 8, args.steps // 5 if not args.stamp_every else args.steps))
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) if i % stamp == 0 else None
           for i in range(args.steps)]
    for pair in evs:
        if pair is not None:
            pair[0].record(stream)
            pair[1].record(stream)
    # Device settle (untimed, ahead of the W warm-up steps, reported as `settle`): after an idle period the GPU needs some
    # tens of milliseconds of work to reach its running clock -- the first ~100 launches last 0.090-0.091 ms against 0.087
    # -- and a run of `--steps 20 --warmup 5` would report that ramp instead of the kernel (42.6 M against 45.3 M solves/s
    # at 200 steps, same build, same box).  Nothing of it is inside the timed region; --settle-ms 0 switches it off.
    # (every settle launch is the SAME cold solve as a timed step -- the sets are put back to the cold state before they are
    # used again -- so a kernel trace or a counter pass over this command averages one workload, not a mix with warm starts)
    settle_launches = 0
    value_unsettled = None
    if args.settle_ms > 0:
        def cold_again():
            for b in warm_sets:
                b.states.copy_(base.states)
                b.warm.copy_(base.warm)
            torch.cuda.synchronize()
        if not use_dist and not args.no_others:
            # what the same command read WITHOUT the settle -- W warm-up steps, K timed steps, as rounds 1-3 reported it --,
            # kept beside the headline so that rounds stay comparable (`settle.value_without`); every set is put back to
            # its cold state afterwards
            for i in range(args.warmup):
                b = warm_sets[i % len(warm_sets)]
                solver.solve_device(base.problems, b.states, b.warm, b.commands, velocities=b.vel)
            torch.cuda.synchronize()
            t_u = time.perf_counter()
            for i in range(args.steps):
                b = sets[i]
                solver.solve_device(base.problems, b.states, b.warm, b.commands, velocities=b.vel)
            torch.cuda.synchronize()
            value_unsettled = cfg["batch"] * args.steps / (time.perf_counter() - t_u)
            for b in sets:
                b.states.copy_(base.states)
                b.warm.copy_(base.warm)
            cold_again()
        t_settle = time.perf_counter()
        while (time.perf_counter() - t_settle) * 1e3 < args.settle_ms:
            cold_again()
            for b in warm_sets:
                solver.solve_device(base.problems, b.states, b.warm, b.commands, velocities=b.vel)
                settle_launches += 1
            torch.cuda.synchronize()
        cold_again()
    for i in range(args.warmup):
        b = warm_sets[i % len(warm_sets)]
        solver.solve_device(base.problems, b.states, b.warm, b.commands, velocities=b.vel)
        if use_dist:
            ev = torch.cuda.Event()
            ev.record(stream)
            exchange(i, b, ev)
    drain()
    extra_streams = [torch.cuda.Stream(device=dev) for _ in range(max(0, args.streams - 1))]
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        b = sets[i]
        st_i = None if args.streams <= 1 or i % args.streams == 0 else extra_streams[i % args.streams - 1].cuda_stream
        solver.solve_device(base.problems, b.states, b.warm, b.commands, velocities=b.vel, events=evs[i], stream=st_i)
        if use_dist:
            exchange(i, b, evs[i][1], timed=True)
    drain()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    kernel_ms = [pair[0].elapsed_time(pair[1]) for pair in evs if pair is not None]

    t = torch.tensor([elapsed], dtype=torch.float64, device=comm_dev)
    per_rank = None
    gather_check = None
    if use_dist:
        # every rank's own time beside the maximum: a SCALE run that falls short shows which rank (GPU) was slow; and a
        # checksum of the commands the rank itself produced in the last step, to hold against its slice of the gathered
        # buffer (rank 0 sees every slice: gathered == the concatenation of the ranks' own commands)
        last = sets[args.steps - 1].vel
        own_sum, own_abs = float(last.sum().item()), float(last.abs().sum().item())
        mine = torch.tensor([elapsed, float(np.mean(kernel_ms)), own_sum, own_abs], dtype=torch.float64, device=comm_dev)
        every = [torch.zeros_like(mine) for _ in range(dist.get_world_size())]
        dist.all_gather(every, mine)
        per_rank = [(float(e[0]), float(e[1])) for e in every]
        g = gathered[(args.steps - 1) % ring]
        ok = [bool(float(g[r].sum().item()) == float(e[2]) and float(g[r].abs().sum().item()) == float(e[3]))
              for r, e in enumerate(every)]
        gather_check = {"ok": all(ok) and bool((g[rank] == last).all().item()), "per_rank": ok,
                        "distinct_slices": len({float(e[3]) for e in every}) == len(every),
                        "what": "sum and sum|.| of every rank's own last-step commands (sent through a second collective) "
                                "against its slice of the gathered buffer on rank 0; rank 0's own slice element by element"}
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())

    cmds = sets[0].commands_host()
    if rank == 0:
        total_instances = cfg["batch"] * world
        value = total_instances * args.steps / elapsed
        k_ms = float(np.mean(kernel_ms))
        roof, pmc = roofline(args.workload, cfg["batch"], n, k_ms, launches_timed=len(kernel_ms))
        out = {
            "metric": "MPC solves/sec (control_steps=%d, %dx%d costmap)" % (n, cfg["map_size"], cfg["map_size"]),
            "value": value, "unit": "solves/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "%s: batch %d instances/GPU, control_steps=%d, horizon=0.8 s, %dx%d u8 costmap, "
                                   "README params, cold start" % (args.workload, cfg["batch"], n, cfg["map_size"],
                                                                  cfg["map_size"]),
                       "parallelism": "instances sharded x%d, 1 RCCL all-gather of (vx,vy,w)/step" % world
                       if world > 1 else "single GPU"},
            "roofline": roof,
            "valu_issue": valu_issue(pmc, k_ms),
            "settle": {"ms": args.settle_ms, "launches": settle_launches, "value_without": value_unsettled,
                       "what": "untimed launches ahead of the W warm-up steps: the clock ramp after idle (--settle-ms 0: off); "
                               "value_without = W warm-up and K timed steps BEFORE the settle, "
                               "as rounds 1-3 reported the metric"},
            **({"study_streams": args.streams} if args.streams > 1 else {}),
            "solver": {"mean_iterations": float(cmds["iterations"].mean()),
                       "max_iterations_seen": int(cmds["iterations"].max()),
                       "converged_frac": float((cmds["status"] == 0).mean()),
                       # status 1 = iteration cap reached (the reference's x.success False, py:399-400)
                       "status_max_iter": int((cmds["status"] == 1).sum())},
        }
        if use_dist:
            out["rccl"] = {"world_size": dist.get_world_size(), "backend": dist.get_backend(), "shared_device": share,
                           "gather_check": gather_check,
                           "gather_ms": float(np.mean([a.elapsed_time(b) for a, b in gather_evs])) if gather_evs else None,
                           "gather_bytes_per_rank": cfg["batch"] * 24,
                           "per_rank": [{"rank": r, "ms_per_step": 1e3 * e / args.steps, "kernel_ms": k,
                                         "value": cfg["batch"] * args.steps / e} for r, (e, k) in enumerate(per_rank)],
                           "what": "one all-gather of (vx, vy, w) per step on a side stream, overlapped with the next "
                                   "step's solve; gather_ms = its own duration (events on the side stream)"}
        if world == 1 and args.streams == 1 and not args.no_others:
            # Two independent batches in flight (NOT the headline, which is one stream, one batch after the other): the same
            # K launches again, alternating over two streams -- a launch of 4096 instances is one residency round whose last
            # third runs on half-empty SIMDs, and the next batch's waves fill them (two fleets served by one GPU).
            try:
                s2 = torch.cuda.Stream(device=dev)
                for b in sets:
                    b.states.copy_(base.states)
                    b.warm.copy_(base.warm)
                torch.cuda.synchronize()
                t2 = time.perf_counter()
                for i, b in enumerate(sets):
                    solver.solve_device(base.problems, b.states, b.warm, b.commands, velocities=b.vel,
                                        stream=s2.cuda_stream if i % 2 else None)
                torch.cuda.synchronize()
                e2 = time.perf_counter() - t2
                out["two_streams"] = {"value": cfg["batch"] * args.steps / e2, "unit": "solves/s", "ms_per_step": 1e3 * e2 / args.steps,
                                      "what": "the same %d cold launches alternating over two streams (two independent fleets); "
                                              "study figure, not `value`" % args.steps}
            except Exception as e:
                out["two_streams"] = {"error": str(e)}
        if world == 1 and not args.no_pcie:
            out["pcie_inclusive"] = pcie_inclusive(solver, probs, st, warm, n)
            # SURVEY 8d's own metric (H2D of the requests and D2H of the results inside the clock), first class beside
            # `value` (the resident-data variant): page-locked arrays worked on in place, and pageable ones
            pin = out["pcie_inclusive"].get("pinned") or {}
            out["value_pcie_inclusive"] = pin.get("value")
            out["value_pcie_inclusive_pageable"] = out["pcie_inclusive"]["value"]
        if world == 1 and not args.no_others and args.workload == "C2" and not args.batch:
            # the other BASELINE configs and the deployed (closed-loop, warm-started) mode, a few launches each, so
            # that the driver's line carries them; the inputs are generated here, outside every timed region
            others = []
            # ("C2/dense direction only": method = NEWTON -- round 5's AUTO at control_steps 3, the A/B partner of the
            # routed headline kernel: no wall model, 5 objective misses in 7332 random-parameter costmap cases against the reference)
            for name, over, label in [("C3", None, None), ("C5", None, None), ("C4", None, "C4 per-GPU shard")] + \
                    [("C2", GENERAL_SETS[k], k) for k in sorted(GENERAL_SETS)] + [("C2", dict(method=2), "C2/dense direction only")]:
                try:
                    # (4096-instance launches last 0.1-0.3 ms: a dozen of them, so that the wall-clock rate is the stream's)
                    others.append(other_workload(name, dev, local_rank, steps=12 if over else 8, params_over=over, label=label))
                except Exception as e:
                    others.append({"workload": label or name, "error": str(e)})
            out["other_workloads"] = others
            try:
                out["warm_tick"] = warm_tick(dev, local_rank)
            except Exception as e:
                out["warm_tick"] = {"error": str(e)}
        if world == 1 and not args.no_cpu_baseline:
            rate, cnt, secs = cpu_baseline(params, cmap, probs)
            out["cpu_baseline"] = {"value": rate, "unit": "solves/s", "cores": 1, "kind": "port",
                                   "cpu_model": cpu_model(), "host_threads_available": os.cpu_count(),
                                   "usable_cpus": usable_cpus(),
                                   "sample": "first %d of the %d %s instances, SciPy SLSQP ftol=%g on the restated "
                                             "Python objective (oracle/mpc_oracle.py), cold start, %.1f s"
                                             % (cnt, cfg["batch"], args.workload, params["opt_tolerance"], secs)}
            try:
                rate_all, done_all, procs = cpu_baseline_all_cores(args.workload)
                out["cpu_baseline"]["all_cores"] = {
                    "value": rate_all, "unit": "solves/s", "processes": procs, "usable_cpus": usable_cpus(),
                    "sample": "%d solves in %d single-threaded processes, 8 s each, the same SciPy port on slices of the first 4096 instances" % (done_all, procs)}
            except Exception as e:   # (the one-core figure is the stated baseline)
                out["cpu_baseline"]["all_cores"] = {"error": str(e)}
            try:
                sub = min(len(probs), 4096)
                out["cpu_mirror"] = {"value": cpu_mirror_rate(params, cmap, probs[:sub], st[:sub], warm[:sub]),
                                     "unit": "solves/s", "cores": usable_cpus(), "omp_threads": usable_cpus(),
                                     "what": "the build's own algorithm in C with OpenMP (oracle/mpc_oracle.c)"}
            except Exception as e:  # the mirror is informational
                out["cpu_mirror"] = {"error": str(e)}
        result_line = json.dumps(out)
    else:
        result_line = None
    if use_dist:
        dist.destroy_process_group()
    # RCCL prints its banner through C stdio, which (when stdout is a pipe) is flushed only at
    # exit -- after Python's own output.  Flush it first so that the JSON line is the last line.
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    if result_line is not None:
        print(result_line, flush=True)


if __name__ == "__main__":
    main()
