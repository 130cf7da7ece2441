"""CPU oracle: a plain-Python/NumPy float64 RESTATEMENT of the reference's MPC hot path.

TEST INFRASTRUCTURE ONLY.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import this module, and only as the
checker / the timed CPU baseline.  The product path (``neo_mpc_planner2_amd``)
never imports it and fails loudly when the HIP library is missing.

Pinned: this restatement is checked against outputs of the reference itself
(``/root/reference/neo_mpc_planner2/mpc_optimization_server.py`` imported under
``oracle/ros_stubs.py``) through the committed vectors in ``tests/golden/``
(generator: ``oracle/gen_golden.py``).  Two boundaries stay "parity unpinned"
because the reference does not contain them:
  * the costmap lookup (``neo_nav2_py_costmap2D``, un-vendored, un-pinned) -- the
    contract below is this build's own;
  * SciPy's SLSQP: README.md:21 names SciPy 1.6.3, goldens were made with the
    SciPy version recorded inside each fixture.

Every function cites the reference lines (``py:`` =
neo_mpc_planner2/mpc_optimization_server.py) it restates.
"""
import math

import numpy as np

# ------------------------------------------------------------------ config contract
# Defaults declared by the reference node (py:49-75).
PY_DEFAULT_PARAMS = dict(
    acc_x_limit=0.5, acc_y_limit=0.5, acc_theta_limit=0.5,
    min_vel_x=-0.5, min_vel_y=-0.5, min_vel_trans=0.5, min_vel_theta=-0.5,
    max_vel_x=0.5, max_vel_y=0.5, max_vel_trans=0.5, max_vel_theta=0.5,
    w_trans=0.5, w_orient=0.5, w_control=0.5, w_terminal=0.5, w_costmap=0.5,
    w_footprint=2000, waiting_time=3.0, low_pass_gain=0.5, opt_tolerance=1e-5,
    prediction_horizon=0.5, control_steps=3,
)
# README sample block (README.md:53-84) -- the "default params" of BASELINE config 1.
README_PARAMS = dict(
    acc_x_limit=2.5, acc_y_limit=2.5, acc_theta_limit=3.0,
    min_vel_x=-0.7, min_vel_y=-0.7, min_vel_trans=-0.7, min_vel_theta=-0.7,
    max_vel_x=0.7, max_vel_y=0.7, max_vel_trans=0.7, max_vel_theta=0.7,
    w_trans=0.82, w_orient=0.50, w_control=0.05, w_terminal=0.05, w_costmap=0.05,
    w_footprint=0, waiting_time=3.0, low_pass_gain=0.5, opt_tolerance=1e-3,
    prediction_horizon=0.8, control_steps=3,
)


def make_params(base=None, **over):
    p = dict(README_PARAMS if base is None else base)
    p.update(over)
    return p


# ------------------------------------------------------------------ costmap contract
def nav2_occupancy_table():
    """raw nav2 cost (u8) -> occupancy in [-1, 100] (build's contract)."""
    t = np.zeros(256, dtype=np.int64)
    v = np.arange(1, 253)
    t[1:253] = 1 + (97 * (v - 1)) // 251
    t[253], t[254], t[255] = 99, 100, -1
    return t


class Costmap:
    """Costmap lookups the reference delegates to neo_nav2_py_costmap2D
    (call sites py:246-247, 257, 262-263, 332-333, 343).  Contract:
      cells[my, mx] raw nav2 u8, row-major; cost = occupancy(raw)/100.0;
      world->map is floor((w - origin)/resolution); out-of-bounds cells cost 1.0."""

    def __init__(self, cells, resolution, origin_x, origin_y):
        self.cells = np.ascontiguousarray(cells, dtype=np.uint8)
        self.size_y, self.size_x = self.cells.shape
        self.resolution = float(resolution)
        self.origin_x = float(origin_x)
        self.origin_y = float(origin_y)
        self.table = nav2_occupancy_table()

    def world_to_map(self, wx, wy):
        mx = int(math.floor((wx - self.origin_x) / self.resolution))
        my = int(math.floor((wy - self.origin_y) / self.resolution))
        return mx, my

    def cost(self, mx, my):
        if mx < 0 or my < 0 or mx >= self.size_x or my >= self.size_y:
            return 1.0
        return float(self.table[self.cells[my, mx]]) / 100.0

    def footprint_cost(self, pts):
        """max cell cost along the closed polygon outline, Bresenham between the
        vertices' cells, end points inclusive; empty polygon -> 0.0."""
        n = len(pts)
        if n == 0:
            return 0.0
        cells = [self.world_to_map(px, py) for (px, py) in pts]
        worst = -1.0
        for i in range(n):
            x0, y0 = cells[i]
            x1, y1 = cells[(i + 1) % n]
            dx, dy = abs(x1 - x0), abs(y1 - y0)
            sx = 1 if x1 >= x0 else -1
            sy = 1 if y1 >= y0 else -1
            err = dx - dy
            x, y = x0, y0
            while True:
                c = self.cost(x, y)
                if c > worst:
                    worst = c
                if x == x1 and y == y1:
                    break
                e2 = 2 * err
                if e2 > -dy:
                    err -= dy
                    x += sx
                if e2 < dx:
                    err += dx
                    y += sy
        return worst


# ------------------------------------------------------------------ problem container
class Problem:
    """One Optimizer.srv request (wire contract SURVEY §8b; cpp:240-246)."""
    __slots__ = ("cur_xy", "cur_q", "carrot_xy", "carrot_q", "goal_xyz", "goal_q",
                 "cur_vel", "control_interval", "delta_t", "footprint")

    def __init__(self, cur_xy, cur_q, carrot_xy, carrot_q, goal_xyz, goal_q, cur_vel,
                 control_interval=1.0 / 30.0, delta_t=0.0, footprint=()):
        self.cur_xy = tuple(float(v) for v in cur_xy)
        self.cur_q = tuple(float(v) for v in cur_q)          # x, y, z, w
        self.carrot_xy = tuple(float(v) for v in carrot_xy)
        self.carrot_q = tuple(float(v) for v in carrot_q)
        self.goal_xyz = tuple(float(v) for v in goal_xyz)
        self.goal_q = tuple(float(v) for v in goal_q)
        self.cur_vel = tuple(float(v) for v in cur_vel)      # vx, vy, wz
        self.control_interval = float(control_interval)
        self.delta_t = float(delta_t)                        # wall clock since last call
        self.footprint = tuple((float(a), float(b)) for (a, b) in footprint)

    def goal_key(self):
        return self.goal_xyz + self.goal_q


def yaw_from_quaternion(x, y, z, w):
    """py:176-178 (yaw component of euler_from_quaternion, py:160-180)."""
    t3 = +2.0 * (w * z + x * y)
    t4 = +1.0 - 2.0 * (y * y + z * z)
    return math.atan2(t3, t4)


def quaternion_from_yaw(yaw):
    """py:182-196 with roll = pitch = 0; returns (w, x, y, z) like the reference's q[0..3]."""
    cy, sy = math.cos(yaw * 0.5), math.sin(yaw * 0.5)
    return (cy, 0.0, 0.0, sy)


def f_constraint(u, index, params):
    """py:157-158."""
    vx, vy = u[0 + index * 3], u[1 + index * 3]
    return params["max_vel_trans"] - math.sqrt(vx * vx + vy * vy)


def objective(u, prob, params, costmap, footprint_cost=None):
    """py:204-269.  ``footprint_cost`` (normalised) overrides the polygon evaluation;
    it is constant in u because of the list aliasing at py:226-244:
    the footprint is evaluated UNTRANSFORMED on every step."""
    n_steps = int(params["control_steps"])
    dt = params["prediction_horizon"] / n_steps                      # py:137
    w_trans, w_orient = params["w_trans"], params["w_orient"]
    w_control, w_terminal = params["w_control"], params["w_terminal"]
    w_costmap, w_footprint = params["w_costmap"], params["w_footprint"]

    target_yaw = yaw_from_quaternion(*prob.carrot_q)                  # py:211
    final_yaw = yaw_from_quaternion(*prob.goal_q)                     # py:212
    # py:213 -- current x, y, z with the GOAL's w (reference quirk, preserved)
    odom_yaw = yaw_from_quaternion(prob.cur_q[0], prob.cur_q[1], prob.cur_q[2], prob.goal_q[3])

    if footprint_cost is None:
        footprint_cost = costmap.footprint_cost(prob.footprint)

    cost_total = 0.0
    x = y = z = 0.0
    vcx, vcy, vcz = prob.cur_vel
    cx, cy = prob.carrot_xy
    pos_x, pos_y = prob.cur_xy
    for i in range(n_steps):
        vx, vy, wz = u[0 + 3 * i], u[1 + 3 * i], u[2 + 3 * i]
        z += wz * dt                                                  # py:230
        x += (vx * math.cos(z) * dt - vy * math.sin(z) * dt)          # py:231
        y += (vx * math.sin(z) * dt + vy * math.cos(z) * dt)          # py:232
        odom_yaw += wz * dt                                           # py:234
        pos_x += vx * math.cos(odom_yaw) * dt - vy * math.sin(odom_yaw) * dt   # py:235
        pos_y += vx * math.sin(odom_yaw) * dt + vy * math.cos(odom_yaw) * dt   # py:236

        mx, my = costmap.world_to_map(pos_x, pos_y)                   # py:246
        c = costmap.cost(mx, my)
        costmap_cost = c ** 2                                         # py:247

        ddx, ddy = cx - x, cy - y
        step_dist_error = math.sqrt(ddx * ddx + ddy * ddy)            # py:250
        step_orient_error = target_yaw - z                            # py:251 (not wrapped)
        cost_total += ((w_trans * step_dist_error ** 2) + (w_orient * step_orient_error ** 2)) / n_steps
        e0, e1, e2 = vcx - vx, vcy - vy, vcz - wz
        cost_total += w_control * math.sqrt(e0 * e0 + e1 * e1 + e2 * e2) / n_steps   # py:253-254

        if c == 1.0:                                                  # py:257-260
            cost_total += costmap_cost * 1000 / n_steps
        else:
            cost_total += w_costmap * costmap_cost / n_steps
        if footprint_cost == 1.0:                                     # py:262-263
            cost_total += (footprint_cost ** 2) * w_footprint / n_steps

    gdx, gdy = cx - prob.goal_xyz[0], cy - prob.goal_xyz[1]
    step_dist_error = math.sqrt(gdx * gdx + gdy * gdy)                # py:266 (carrot vs goal)
    step_orient_error = final_yaw - z                                 # py:267
    cost_total += ((w_trans * step_dist_error ** 2) + (w_orient * step_orient_error ** 2)) * w_terminal
    return cost_total


def initial_guess_update(init_guess, guess, n_steps):
    """py:198-202."""
    for i in range(0, n_steps - 1):
        init_guess[0 + 3 * i:3 + 3 * i] = guess[3 + 3 * i:6 + 3 * i]
    init_guess[0 + 3 * (n_steps - 1):3 + 3 * (n_steps - 1)] = guess[0:3]
    return init_guess


def collision_check(x, prob, params, costmap):
    """py:312-341: global-frame rollout (TRUE yaw, py:317); returns True when a
    predicted cell has cost >= 0.99."""
    n_steps = int(params["control_steps"])
    dt = params["prediction_horizon"] / n_steps
    pos_x, pos_y = prob.cur_xy
    odom_yaw = yaw_from_quaternion(*prob.cur_q)
    for i in range(n_steps):
        odom_yaw += x[2 + 3 * i] * dt
        pos_x += x[3 * i] * math.cos(odom_yaw) * dt - x[1 + 3 * i] * math.sin(odom_yaw) * dt
        pos_y += x[3 * i] * math.sin(odom_yaw) * dt + x[1 + 3 * i] * math.cos(odom_yaw) * dt
        mx, my = costmap.world_to_map(pos_x, pos_y)
        if costmap.cost(mx, my) >= 0.99:
            return True
    return False


def predicted_path(x, prob, params):
    """py:293-306 rollout (as published on `local_plan`), started from the request's
    current pose: list of (X, Y, yaw)."""
    n_steps = int(params["control_steps"])
    dt = params["prediction_horizon"] / n_steps
    pos_x, pos_y = prob.cur_xy
    yaw = yaw_from_quaternion(*prob.cur_q)
    out = []
    for i in range(n_steps):
        yaw += x[2 + 3 * i] * dt
        pos_x += x[3 * i] * math.cos(yaw) * dt - x[1 + 3 * i] * math.sin(yaw) * dt
        pos_y += x[3 * i] * math.sin(yaw) * dt + x[1 + 3 * i] * math.cos(yaw) * dt
        out.append((pos_x, pos_y, yaw))
    return out


def bounds_and_constraints(params):
    """py:125-134."""
    n_steps = int(params["control_steps"])
    bnds, cons = [], []
    for i in range(n_steps):
        bnds.append((params["min_vel_x"], params["max_vel_x"]))
        bnds.append((params["min_vel_y"], params["max_vel_y"]))
        bnds.append((params["min_vel_theta"], params["max_vel_theta"]))
        cons.append({"type": "ineq", "fun": (lambda u, i=i: f_constraint(u, i, params))})
    return bnds, cons


def solve_slsqp(prob, params, costmap, x0, ftol=None, maxiter=100, footprint_cost=None):
    """py:363-364: the SciPy call with the restated objective.  Returns the SciPy result."""
    from scipy.optimize import minimize
    if footprint_cost is None:
        footprint_cost = costmap.footprint_cost(prob.footprint)
    bnds, cons = bounds_and_constraints(params)
    return minimize(lambda u: objective(u, prob, params, costmap, footprint_cost),
                    np.array(x0, dtype=np.float64), method="SLSQP", bounds=bnds, constraints=cons,
                    options={"ftol": params["opt_tolerance"] if ftol is None else ftol,
                             "disp": False, "maxiter": maxiter})


class ServerState:
    """Per-instance persistent state of the reference node (py:115-152)."""

    def __init__(self, n_steps):
        self.initial_guess = np.zeros(3 * n_steps)
        self.last_control = [0.0, 0.0, 0.0]
        self.old_goal = None           # py:146: a PoseStamped never equals a Pose -> first call resets
        self.collision = False
        self.collision_footprint = False
        self.waiting_time = 3.0        # py:103 (param), clobbered at py:361


def optimizer_step(state, prob, params, costmap, solver=None):
    """py:349-403.  ``solver(prob, params, costmap, x0, footprint_cost) -> (x, success)``;
    default = SciPy SLSQP at ``opt_tolerance``.  Returns (vx, vy, wz) and mutates
    ``state``.  ``prob.delta_t`` stands in for the wall-clock difference of py:369-371."""
    n_steps = int(params["control_steps"])
    if state.old_goal is None or state.old_goal != prob.goal_key():      # py:358-361
        state.initial_guess = np.zeros(3 * n_steps)
        state.last_control = [0.0, 0.0, 0.0]
        state.waiting_time = 0.0

    footprint_cost = costmap.footprint_cost(prob.footprint)
    if solver is None:
        res = solve_slsqp(prob, params, costmap, state.initial_guess, footprint_cost=footprint_cost)
        x, success = np.array(res.x, dtype=np.float64), bool(res.success)
    else:
        x, success = solver(prob, params, costmap, state.initial_guess.copy(), footprint_cost)
        x = np.array(x, dtype=np.float64)

    g = params["low_pass_gain"]
    for i in range(3):                                                     # py:366-367
        x[i] = x[i] * g + state.last_control[i] * (1 - g)

    if collision_check(x, prob, params, costmap):                          # py:338-341 (latch)
        state.collision = True
    state.collision_footprint = (footprint_cost == 1.0)                    # py:343-347

    if state.collision or state.collision_footprint:                       # py:374-382
        out = (0.0, 0.0, 0.0)
        state.waiting_time += prob.delta_t
        if state.waiting_time >= 3.0:
            state.collision = False
            state.waiting_time = 0.0
    else:                                                                  # py:383-391
        ci = prob.control_interval
        acc = (params["acc_x_limit"], params["acc_y_limit"], params["acc_theta_limit"])
        out = []
        for i in range(3):
            t = float(np.fmin(x[i], state.last_control[i] + acc[i] * ci))
            out.append(float(np.fmax(t, state.last_control[i] - acc[i] * ci)))
        out = tuple(out)

    state.last_control = [out[0], out[1], out[2]]                          # py:393-395
    if success:                                                            # py:397-400
        state.initial_guess = initial_guess_update(state.initial_guess, x, n_steps)
    else:
        state.initial_guess = x
    state.old_goal = prob.goal_key()                                       # py:402
    return out, x
