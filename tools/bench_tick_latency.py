#!/usr/bin/env python3
"""Single-robot control tick through the reference's own surface (BASELINE config 1: one MPO-700
problem, control_steps=3, horizon 0.8 s, 200x200 costmap, README params): per tick the costmap is
handed over (`set_costmap`, as the plugin does with costmap_->getCharMap()) and
`MpcOptimizationServer.optimizer(request, response)` is called.  Reference: 11.6 ms per solve on
one core, plus the DDS service hop."""
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from neo_mpc_planner2_amd import mpc_optimization_server as srv, synthetic  # noqa: E402

cmap = synthetic.make_costmap(200, seed=0)
node = srv.MpcOptimizationServer(srv.README_PARAMS)
row = synthetic.make_problems(1, 200, seed=3)[0]
pos, yaw, vel = np.array(row["cur_xy"]), 0.2, np.zeros(3)
lat_solve, lat_map = [], []
for k in range(300):
    req = srv.make_request(pos, synthetic.yaw_quat(np.array(yaw)), row["carrot_xy"], row["carrot_q"],
                           row["goal_xyz"], row["goal_q"], vel, control_interval=1 / 30)
    t0 = time.perf_counter()
    node.set_costmap(*cmap)
    t1 = time.perf_counter()
    resp = node.optimizer(req, srv.make_response())
    t2 = time.perf_counter()
    if k >= 20:
        lat_map.append(t1 - t0)
        lat_solve.append(t2 - t1)
    vel = np.array([resp.output_vel.twist.linear.x, resp.output_vel.twist.linear.y, resp.output_vel.twist.angular.z])
    yaw += vel[2] / 30
    pos = pos + np.array([vel[0] * math.cos(yaw) - vel[1] * math.sin(yaw),
                          vel[0] * math.sin(yaw) + vel[1] * math.cos(yaw)]) / 30
print(json.dumps({"config": "C1: single robot tick, 200x200 costmap, warm-started episode",
                  "optimizer_call_ms_median": 1e3 * float(np.median(lat_solve)),
                  "optimizer_call_ms_p99": 1e3 * float(np.percentile(lat_solve, 99)),
                  "set_costmap_ms_median": 1e3 * float(np.median(lat_map)),
                  "ticks_per_s": 1.0 / float(np.mean(lat_solve) + np.mean(lat_map))}))
