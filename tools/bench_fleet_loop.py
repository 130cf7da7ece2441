#!/usr/bin/env python3
"""Closed control loop of a fleet on one GPU: 4096 robots (the C2 problem), state resident in HBM,
one K1 launch per tick at 30 Hz simulated time; between ticks the robots are moved by their own
commands and the carrot -- kept 0.4 m ahead of the robot along its initial world bearing, as a
look-ahead point sliding along a straight plan would be -- is re-expressed in the base frame, all on
the device (torch ops on the request records).  Tick 1 is the cold start the headline benchmark
measures; from tick 2 on the solver is warm-started the reference's way (py:397-400: the previous
solution shifted by one whole control step of 0.267 s although only 1/30 s has passed), so a warm
tick needs about as many iterations as a cold one."""
import json
import math
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from neo_mpc_planner2_amd import abi, fleet, synthetic  # noqa: E402
from neo_mpc_planner2_amd.mpc_optimization_server import README_PARAMS  # noqa: E402
from neo_mpc_planner2_amd.solver import BatchSolver, DeviceBatch  # noqa: E402

TICKS, HZ = int(os.environ.get("NEO_MPC_TICKS", "60")), 30.0
DUMP_TICK = int(os.environ.get("NEO_MPC_DUMP_TICK", "-1"))
POOL = "--pool" in sys.argv   # every robot gets its own 200x200 rolling window (neo_mpc_set_costmap_pool)
cfg, cmap, probs, st, warm = synthetic.make_workload("C2", seed=0)
params = dict(README_PARAMS)
params.update(control_steps=3)
if os.environ.get("NEO_MPC_KINK_RADIUS"):   # study knob
    params["kink_radius"] = float(os.environ["NEO_MPC_KINK_RADIUS"])
dev = "cuda:0"
with BatchSolver(params) as s:
    if POOL:
        # 64 distinct synthetic windows, repeated; window k is centred on robot k and re-centred (origins
        # rewritten on the device) every tick, its contents re-ingested (K3) every tick like a fresh costmap
        base = np.stack([synthetic.make_costmap(200, seed=100 + k)[0] for k in range(64)])
        d_cells = torch.from_numpy(base).to(dev).repeat(len(probs) // 64, 1, 1).contiguous()
        probs["map_index"] = np.arange(len(probs), dtype=np.int32)
        d_orig = torch.from_numpy(np.ascontiguousarray(probs["cur_xy"]) - 5.0).to(dev)
        s.set_costmap_pool(d_cells, 0.05, d_orig)
    else:
        s.set_costmap(torch.from_numpy(cmap[0]).to(dev), *cmap[1:])
    b = DeviceBatch(probs, st, warm, dev, want_solution=False)
    ing = []

    def before_tick(t, pos):
        if POOL:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            d_orig.copy_(pos - 5.0)
            s.set_costmap_pool(d_cells, 0.05, d_orig)
            e1.record()
            ing.append((e0, e1))
        if DUMP_TICK == t:   # study: the inputs of this tick, for the CPU mirror (tools: trace the long searches)
            np.savez(os.path.join(ROOT, "gpurun_out", "fleet_tick%d.npz" % t), problems=b.problems.cpu().numpy(),
                     states=b.states.cpu().numpy(), warm=b.warm.cpu().numpy())

    def after_tick(t, cm):
        if DUMP_TICK == t:
            np.save(os.path.join(ROOT, "gpurun_out", "fleet_tick%d_iterations.npy" % t), cm["iterations"])

    loop = fleet.closed_loop(s, b, TICKS, HZ, before_tick, after_tick)
    ms, iters, itmax, stopped = loop["kernel_ms"], loop["mean_iterations"], loop["max_iterations"], loop["stopped_fraction"]
extra = {}
if POOL:
    extra = {"pool": "4096 rolling windows of 200x200 cells (160 MB raw), re-centred and re-ingested every tick",
             "ingest_ms_median": float(np.median([a.elapsed_time(e) for a, e in ing[5:]]))}
print(json.dumps({
    **extra,
    "config": "C2 fleet in closed loop: 4096 robots, control_steps=3, %s, 30 Hz, state resident"
              % ("one 200x200 costmap per robot" if POOL else "500x500 map"),
    "tick1_cold_kernel_ms": ms[0], "tick1_mean_iterations": iters[0],
    "warm_ticks_kernel_ms_median": float(np.median(ms[5:])), "warm_ticks_kernel_ms_max": float(np.max(ms[5:])),
    "warm_ticks_mean_iterations": float(np.mean(iters[5:])),
    "warm_solves_per_s": 4096 / (1e-3 * float(np.median(ms[5:]))),
    "stopped_fraction_last_tick": stopped[-1],
    "per_tick_kernel_ms": [round(x, 4) for x in ms[:12]], "per_tick_mean_iterations": [round(x, 2) for x in iters[:12]],
    "per_tick_max_iterations": itmax[:12], "warm_ticks_max_iterations_median": float(np.median(itmax[5:])),
    "warm_ticks_max_iterations_max": int(np.max(itmax[5:])), "ticks": TICKS}))
