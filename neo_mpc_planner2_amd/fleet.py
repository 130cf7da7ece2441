"""Closed control loop of a fleet on one GPU -- the deployed mode of the hot path.

The reference runs one robot: `computeVelocityCommands` (src/NeoMpcPlanner.cpp:202-254) at `controller_frequency`,
each call warm-started from the previous solution shifted by one control step (py:397-400, 198-202).  Here a whole
fleet does that on the device: one K1 launch per tick over HBM-resident requests / state / warm starts; between ticks
the robots are moved by their own commands and the look-ahead point -- kept 0.4 m ahead of the robot along its initial
world bearing, as a carrot sliding along a straight plan would be -- is re-expressed in the base frame with torch ops
on the request records (plumbing: the tick itself goes through the C-ABI).  Used by bench.py (`warm_tick`) and
tools/bench_fleet_loop.py.
"""
import numpy as np

from . import abi


def closed_loop(solver, batch, ticks, hz=30.0, before_tick=None, after_tick=None, balance_every=0):
    """Run `ticks` control ticks of `batch` (a solver.DeviceBatch) through `solver` (a BatchSolver with its costmap
    set).  Returns per-tick lists: kernel_ms (HIP events around the K1 launch), mean_iterations, max_iterations,
    stopped_fraction.  `before_tick(t, pos)` runs before tick t's launch (e.g. re-centre a costmap pool),
    `after_tick(t, commands)` after its commands are on the host.  `balance_every` = R > 0: every R-th tick the
    dispatch order of the following ticks is rebuilt from that tick's iteration counts (neo_mpc_balance_dispatch_device:
    one small kernel behind K1 on the same stream; `balance_ms` in the result is its own duration) -- results do not
    depend on it, only which robots share a SIMD."""
    import torch
    b = batch
    P = b.problems.view(torch.float64).reshape(b.count, -1)          # the 32 doubles of each request
    q = P[:, 2:6]
    yaw = torch.atan2(2 * (q[:, 3] * q[:, 2] + q[:, 0] * q[:, 1]), 1 - 2 * (q[:, 1] ** 2 + q[:, 2] ** 2)).clone()
    pos = P[:, 0:2].clone()
    c, sn = torch.cos(yaw), torch.sin(yaw)
    carrot_off = torch.stack([c * P[:, 6] - sn * P[:, 7], sn * P[:, 6] + c * P[:, 7]], 1)   # world frame
    cq = P[:, 8:12]
    carrot_yaw_w = yaw + torch.atan2(2 * (cq[:, 3] * cq[:, 2] + cq[:, 0] * cq[:, 1]),
                                     1 - 2 * (cq[:, 1] ** 2 + cq[:, 2] ** 2))
    P[:, 22] = 1.0 / hz
    P[:, 23] = 1.0 / hz
    # one event pair per tick, stamped by the dispatch of K1 itself (neo_mpc_solve_batch_device_timed): separate
    # event records around the launch would add their own barrier packets (~20 us) to what is read as kernel time
    stream = torch.cuda.current_stream()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(ticks)]
    for e0, e1 in evs:
        e0.record(stream)
        e1.record(stream)
    torch.cuda.synchronize()
    out = {"kernel_ms": [], "mean_iterations": [], "max_iterations": [], "stopped_fraction": []}
    bal_evs = []
    for t in range(ticks):
        if before_tick is not None:
            before_tick(t, pos)
        solver.solve_device(b.problems, b.states, b.warm, b.commands, velocities=b.vel, events=evs[t])
        if balance_every and t % balance_every == 0:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            solver.balance_dispatch(b.commands)
            e1.record(stream)
            bal_evs.append((e0, e1))
        cmd = b.vel
        yaw = yaw + cmd[:, 2] / hz
        c, sn = torch.cos(yaw), torch.sin(yaw)
        pos = pos + torch.stack([c * cmd[:, 0] - sn * cmd[:, 1], sn * cmd[:, 0] + c * cmd[:, 1]], 1) / hz
        P[:, 0:2] = pos
        P[:, 2] = 0.0
        P[:, 3] = 0.0
        P[:, 4] = torch.sin(0.5 * yaw)
        P[:, 5] = torch.cos(0.5 * yaw)
        d = carrot_off
        P[:, 6] = c * d[:, 0] + sn * d[:, 1]
        P[:, 7] = -sn * d[:, 0] + c * d[:, 1]
        rel = carrot_yaw_w - yaw
        P[:, 8] = 0.0
        P[:, 9] = 0.0
        P[:, 10] = torch.sin(0.5 * rel)
        P[:, 11] = torch.cos(0.5 * rel)
        P[:, 19:22] = cmd
        torch.cuda.synchronize()
        cm = b.commands.cpu().numpy().view(abi.COMMAND_DTYPE).reshape(-1)
        out["mean_iterations"].append(float(cm["iterations"].mean()))
        out["max_iterations"].append(int(cm["iterations"].max()))
        out["stopped_fraction"].append(float(((cm["flags"] & abi.FLAG_STOPPED) != 0).mean()))
        if after_tick is not None:
            after_tick(t, cm)
    out["kernel_ms"] = [a.elapsed_time(e) for a, e in evs]
    if bal_evs:
        out["balance_ms"] = [a.elapsed_time(e) for a, e in bal_evs]
        out["balance_every"] = balance_every
        solver.balance_dispatch(None)
    return out


def summary(loop, skip=5):
    """The figures bench.py reports for the warm ticks (everything after the first `skip`)."""
    ms, it, mx = loop["kernel_ms"], loop["mean_iterations"], loop["max_iterations"]
    return {"ticks": len(ms) - skip, "ms_per_tick_median": float(np.median(ms[skip:])),
            "ms_per_tick_max": float(np.max(ms[skip:])), "mean_iterations": float(np.mean(it[skip:])),
            "max_iterations_median": float(np.median(mx[skip:])), "max_iterations_max": int(np.max(mx[skip:])),
            "cold_tick_ms": float(ms[0]), "cold_tick_mean_iterations": float(it[0]),
            "stopped_fraction_last_tick": float(loop["stopped_fraction"][-1]),
            # balanced dispatch (closed_loop(balance_every=R)): the order kernel's own duration (events around it, launch
            # gaps included) and its share of a tick when it runs every R-th tick
            **({"balance_every": loop["balance_every"], "order_kernel_ms_median": float(np.median(loop["balance_ms"])),
                "ms_per_tick_median_incl_order": float(np.median(ms[skip:]) + np.median(loop["balance_ms"]) / loop["balance_every"])}
               if "balance_ms" in loop else {})}
