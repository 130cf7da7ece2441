"""Seeded synthetic workloads for the MPC hot path (SURVEY.md §8d "Concrete synthetic
inputs"): a shared nav2-style costmap and per-instance Optimizer requests.

NumPy only; used by bench.py, the tests and the golden-vector generator.
"""
import math

import numpy as np

from .abi import PROBLEM_DTYPE, STATE_DTYPE

RESOLUTION = 0.05

#: BASELINE.json configs (batch, control_steps, costmap side)
CONFIGS = {
    "C1": dict(batch=1, control_steps=3, map_size=200),
    "C2": dict(batch=4096, control_steps=3, map_size=500),
    "C3": dict(batch=262144, control_steps=8, map_size=1000),
    "C4": dict(batch=2097152, control_steps=3, map_size=500),   # sharded over 8 GPUs
    "C5": dict(batch=65536, control_steps=32, map_size=500),
}


def make_costmap(size, seed=0, resolution=RESOLUTION, n_discs=None):
    """size x size raw nav2 costmap (u8): K = size/25 lethal discs (254) of radius
    U[0.2, 0.6] m, inscribed (253) within 0.3 m of them, then nav2's exponential
    inflation 252*exp(-3.0*(d - 0.3)), free (0) elsewhere.
    Returns (cells[size, size], resolution, origin_x, origin_y)."""
    rng = np.random.default_rng(seed)
    origin = -size * resolution / 2.0
    k = max(1, size // 25) if n_discs is None else n_discs
    centres = rng.uniform(origin, -origin, size=(k, 2))
    radii = rng.uniform(0.2, 0.6, size=k)
    axis = origin + (np.arange(size) + 0.5) * resolution
    dist = np.full((size, size), np.inf)
    for (cx, cy), r in zip(centres, radii):
        d = np.sqrt((axis[None, :] - cx) ** 2 + (axis[:, None] - cy) ** 2) - r
        np.minimum(dist, d, out=dist)
    cells = np.zeros((size, size), dtype=np.uint8)
    infl = np.floor(252.0 * np.exp(-3.0 * (np.maximum(dist, 0.3) - 0.3)))
    infl[dist > 1.5] = 0
    cells[:] = infl.astype(np.uint8)
    cells[dist <= 0.3] = 253
    cells[dist <= 0.0] = 254
    return cells, float(resolution), float(origin), float(origin)


def yaw_quat(yaw):
    """planar quaternion (x, y, z, w) arrays for yaw arrays."""
    yaw = np.asarray(yaw, dtype=np.float64)
    q = np.zeros(yaw.shape + (4,), dtype=np.float64)
    q[..., 2] = np.sin(0.5 * yaw)
    q[..., 3] = np.cos(0.5 * yaw)
    return q


def make_problems(count, map_size, seed=0, resolution=RESOLUTION, control_interval=1.0 / 30.0,
                  same_goal_w=False):
    """`count` independent requests drawn as SURVEY §8d specifies:
    position U over the map interior (>= 1 m from the border), yaw U[-pi, pi];
    carrot at 0.4 m, bearing U[-pi, pi], yaw U[-1.5, 1.5] (base frame);
    goal U over the map, yaw U[-pi, pi]; current velocity U[-0.5, 0.5]^3."""
    rng = np.random.default_rng(seed)
    half = map_size * resolution / 2.0
    p = np.zeros(count, dtype=PROBLEM_DTYPE)
    p["cur_xy"] = rng.uniform(-half + 1.0, half - 1.0, size=(count, 2))
    cur_yaw = rng.uniform(-math.pi, math.pi, size=count)
    p["cur_q"] = yaw_quat(cur_yaw)
    bearing = rng.uniform(-math.pi, math.pi, size=count)
    p["carrot_xy"][:, 0] = 0.4 * np.cos(bearing)
    p["carrot_xy"][:, 1] = 0.4 * np.sin(bearing)
    p["carrot_q"] = yaw_quat(rng.uniform(-1.5, 1.5, size=count))
    p["goal_xyz"][:, :2] = rng.uniform(-half, half, size=(count, 2))
    goal_yaw = cur_yaw if same_goal_w else rng.uniform(-math.pi, math.pi, size=count)
    p["goal_q"] = yaw_quat(goal_yaw)
    p["cur_vel"] = rng.uniform(-0.5, 0.5, size=(count, 3))
    p["control_interval"] = control_interval
    p["delta_t"] = control_interval
    p["footprint_cost"] = 0.0
    return p


def make_states(problems, control_steps):
    """State for a timed tick in the middle of an episode: the goal is unchanged (no
    reset), `last_control` = current velocity, cold warm start."""
    count = problems.shape[0]
    st = np.zeros(count, dtype=STATE_DTYPE)
    st["last_control"] = problems["cur_vel"]
    st["old_goal"][:, :3] = problems["goal_xyz"]
    st["old_goal"][:, 3:] = problems["goal_q"]
    st["has_old_goal"] = 1
    st["waiting_time"] = 0.0
    warm = np.zeros((count, 3 * control_steps), dtype=np.float64)
    return st, warm


def make_workload(name, seed=0, batch=None):
    """(params overrides, costmap tuple, problems, states, warm) for a BASELINE config."""
    cfg = dict(CONFIGS[name])
    if batch is not None:
        cfg["batch"] = batch
    cmap = make_costmap(cfg["map_size"], seed=seed)
    probs = make_problems(cfg["batch"], cfg["map_size"], seed=seed + 1000)
    st, warm = make_states(probs, cfg["control_steps"])
    return cfg, cmap, probs, st, warm


#: a rectangular robot outline in the base frame (metres)
RECT_FOOTPRINT = ((0.35, 0.25), (-0.35, 0.25), (-0.35, -0.25), (0.35, -0.25))


def footprint_world(prob_row, base=RECT_FOOTPRINT):
    """The footprint polygon nav2 publishes on /local_costmap/published_footprint: `base`
    placed at the request's current pose (global frame)."""
    x0, y0 = prob_row["cur_xy"]
    q = prob_row["cur_q"]
    yaw = math.atan2(2.0 * (q[3] * q[2] + q[0] * q[1]), 1.0 - 2.0 * (q[1] * q[1] + q[2] * q[2]))
    c, s = math.cos(yaw), math.sin(yaw)
    return [(x0 + px * c - py * s, y0 + px * s + py * c) for (px, py) in base]


def make_plans(count, seed=0, min_len=64, max_len=512, spacing=0.05):
    """Ragged batch of global plans for the carrot selection (NeoMpcPlanner.cpp:66-189): smooth
    random polylines of `min_len..max_len` poses `spacing` metres apart with heading = direction
    of travel, and a robot pose near a random pose of each plan.
    Returns (plan_poses[total, 3], plan_offsets[count+1] uint32, robot_poses[count, 3])."""
    rng = np.random.default_rng(seed)
    lens = rng.integers(min_len, max_len + 1, size=count)
    offsets = np.zeros(count + 1, dtype=np.uint32)
    offsets[1:] = np.cumsum(lens)
    poses = np.zeros((int(offsets[-1]), 3), dtype=np.float64)
    robots = np.zeros((count, 3), dtype=np.float64)
    for i in range(count):
        n = int(lens[i])
        heading = rng.uniform(-math.pi, math.pi) + np.cumsum(rng.normal(0.0, 0.03, size=n))
        xy = rng.uniform(-10, 10, size=2) + spacing * np.cumsum(np.stack([np.cos(heading), np.sin(heading)], 1), axis=0)
        poses[offsets[i]:offsets[i + 1], :2] = xy
        poses[offsets[i]:offsets[i + 1], 2] = heading
        k = int(rng.integers(0, n))
        robots[i, :2] = xy[k] + rng.normal(0.0, 0.05, size=2)
        robots[i, 2] = heading[k] + rng.uniform(-1.6, 1.6)
    return poses, offsets, robots
