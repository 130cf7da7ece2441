"""Host-side mirror of the reference's service node, backed by the HIP solver.

`MpcOptimizationServer` keeps the public surface of
neo_mpc_planner2/mpc_optimization_server.py (class name, ROS parameter names and
defaults py:49-75, `optimizer(request, response)` py:349-403, `footprint_callback`
py:154-155, `cb_params` py:405-439, `initial_guess` / `last_control` / `collision`
attributes) so that code and tests written against the reference read the same, but the
SciPy `minimize` call and everything after it run in one HIP kernel launch through the
C-ABI.  No ROS imports: requests / responses are duck-typed (any object with the
`neo_srvs2/srv/Optimizer` attribute tree works, including real rclpy messages).
"""
import time
import types

import numpy as np

from . import abi
from .solver import BatchSolver

#: defaults declared by the reference node (py:49-75)
DEFAULT_PARAMS = dict(
    acc_x_limit=0.5, acc_y_limit=0.5, acc_theta_limit=0.5,
    min_vel_x=-0.5, min_vel_y=-0.5, min_vel_trans=0.5, min_vel_theta=-0.5,
    max_vel_x=0.5, max_vel_y=0.5, max_vel_trans=0.5, max_vel_theta=0.5,
    w_trans=0.5, w_orient=0.5, w_control=0.5, w_terminal=0.5, w_costmap=0.5,
    w_footprint=2000, waiting_time=3.0, low_pass_gain=0.5, opt_tolerance=1e-5,
    prediction_horizon=0.5, control_steps=3,
)
#: the README sample block (README.md:53-84), BASELINE's "default params"
README_PARAMS = dict(
    acc_x_limit=2.5, acc_y_limit=2.5, acc_theta_limit=3.0,
    min_vel_x=-0.7, min_vel_y=-0.7, min_vel_trans=-0.7, min_vel_theta=-0.7,
    max_vel_x=0.7, max_vel_y=0.7, max_vel_trans=0.7, max_vel_theta=0.7,
    w_trans=0.82, w_orient=0.50, w_control=0.05, w_terminal=0.05, w_costmap=0.05,
    w_footprint=0, waiting_time=3.0, low_pass_gain=0.5, opt_tolerance=1e-3,
    prediction_horizon=0.8, control_steps=3,
)
#: parameters `cb_params` accepts at run time (py:408-435)
DYNAMIC_PARAMS = ("min_vel_x", "min_vel_y", "min_vel_trans", "min_vel_theta", "max_vel_x", "max_vel_y",
                  "max_vel_trans", "max_vel_theta", "w_trans", "w_orient", "w_control", "w_terminal",
                  "w_costmap", "w_footprint")
#: ... of which only these change the reference's behaviour: the velocity box is baked into
#: `self.bnds` at construction (py:125-133, never rebuilt), `min_vel_trans` is unused, and the
#: w_costmap / w_footprint branches assign attributes nothing reads (py:433, 435; the objective
#: uses `w_costmap_scale` / `w_footprint_scale`, py:260, 263).  SURVEY.md Appendix A #12, #13.
EFFECTIVE_DYNAMIC_PARAMS = ("max_vel_trans", "w_trans", "w_orient", "w_control", "w_terminal")


def load_params_yaml(path, node="mpc_optimization_server"):
    """Read the `<node>: ros__parameters:` block of a nav2 params YAML (README.md:51-84):
    the YAML and the parameter names stay exactly what the reference uses."""
    import yaml
    with open(path) as f:
        doc = yaml.safe_load(f)
    block = doc.get(node, doc)
    block = block.get("ros__parameters", block)
    p = dict(DEFAULT_PARAMS)
    for k, v in block.items():
        if k in p:
            p[k] = v
    return p


def _ns(**kw):
    return types.SimpleNamespace(**kw)


def make_response():
    """An empty `Optimizer.Response`-shaped object (output_vel.twist.linear/angular)."""
    return _ns(output_vel=_ns(header=_ns(stamp=0, frame_id=""),
                              twist=_ns(linear=_ns(x=0.0, y=0.0, z=0.0), angular=_ns(x=0.0, y=0.0, z=0.0))))


def make_request(cur_xy=(0.0, 0.0), cur_q=(0, 0, 0, 1), carrot_xy=(0.0, 0.0), carrot_q=(0, 0, 0, 1),
                 goal_xyz=(0.0, 0.0, 0.0), goal_q=(0, 0, 0, 1), cur_vel=(0.0, 0.0, 0.0),
                 control_interval=1.0 / 30.0, switch_opt=False):
    """An `Optimizer.Request`-shaped object (cpp:240-246)."""
    def pose(xy, q, z=0.0):
        return _ns(position=_ns(x=float(xy[0]), y=float(xy[1]), z=float(z)),
                   orientation=_ns(x=float(q[0]), y=float(q[1]), z=float(q[2]), w=float(q[3])))
    return _ns(current_pose=_ns(header=_ns(), pose=pose(cur_xy, cur_q)),
               carrot_pose=_ns(header=_ns(), pose=pose(carrot_xy, carrot_q)),
               goal_pose=pose(goal_xyz[:2], goal_q, goal_xyz[2] if len(goal_xyz) > 2 else 0.0),
               current_vel=_ns(linear=_ns(x=float(cur_vel[0]), y=float(cur_vel[1]), z=0.0),
                               angular=_ns(x=0.0, y=0.0, z=float(cur_vel[2]))),
               switch_opt=bool(switch_opt), control_interval=float(control_interval))


def request_record(request, delta_t=0.0, footprint_cost=0.0):
    """`Optimizer.Request` -> one `neo_mpc_problem` record (py:350-355 field for field)."""
    r = np.zeros(1, dtype=abi.PROBLEM_DTYPE)
    cp, co = request.current_pose.pose.position, request.current_pose.pose.orientation
    kp, ko = request.carrot_pose.pose.position, request.carrot_pose.pose.orientation
    gp, go = request.goal_pose.position, request.goal_pose.orientation
    r["cur_xy"] = (cp.x, cp.y)
    r["cur_q"] = (co.x, co.y, co.z, co.w)
    r["carrot_xy"] = (kp.x, kp.y)
    r["carrot_q"] = (ko.x, ko.y, ko.z, ko.w)
    r["goal_xyz"] = (gp.x, gp.y, gp.z)
    r["goal_q"] = (go.x, go.y, go.z, go.w)
    r["cur_vel"] = (request.current_vel.linear.x, request.current_vel.linear.y, request.current_vel.angular.z)
    r["control_interval"] = request.control_interval
    r["delta_t"] = delta_t
    r["footprint_cost"] = footprint_cost
    r["switch_opt"] = 1 if getattr(request, "switch_opt", False) else 0    # cpp:245 (stored py:354, never read)
    return r


class MpcOptimizationServer:
    """Single-robot drop-in for the reference node (py:44): one instance, batch size 1.

    parameters: dict of the reference's ROS parameter names (missing ones take py:49-75
    defaults).  `clock` replaces `time.time` (py:369) for deterministic tests."""

    def __init__(self, parameters=None, device=0, clock=time.time, reference_quirks=True):
        """reference_quirks: reproduce the reference's dynamic-reconfigure behaviour, where only
        EFFECTIVE_DYNAMIC_PARAMS reach the optimisation; False makes every accepted name effective."""
        self.reference_quirks = bool(reference_quirks)
        p = dict(DEFAULT_PARAMS)
        p.update(parameters or {})
        for name, value in p.items():                       # py:78-103
            setattr(self, name, value)
        self.no_ctrl_steps = int(p["control_steps"])         # py:102
        self.w_costmap_scale = p["w_costmap"]                # py:96
        self.w_footprint_scale = p["w_footprint"]            # py:97
        self.dt = p["prediction_horizon"] / self.no_ctrl_steps   # py:137
        self._params = p
        self._clock = clock
        self._solver = BatchSolver(p, device=device)
        self._state, self._warm = abi.new_states(1, self.no_ctrl_steps, waiting_time=p["waiting_time"])
        self.last_time = 0.0                                 # py:138
        self.footprint = None                                # py:154-155
        self.control_interval = 0.0                          # py:152
        self.last_result = None
        self.local_plan = None

    # -- state the reference exposes as attributes -------------------------------------
    @property
    def initial_guess(self):                                 # py:136
        return self._warm[0]

    @property
    def last_control(self):                                  # py:117
        return list(self._state["last_control"][0])

    @property
    def collision(self):                                     # py:148
        return bool(self._state["collision"][0])

    @property
    def collision_footprint(self):                           # py:149
        return bool(self._state["collision_footprint"][0])

    @property
    def waiting_time_state(self):
        return float(self._state["waiting_time"][0])

    # -- inputs that are not part of the request ---------------------------------------
    def set_costmap(self, cells, resolution, origin_x, origin_y):
        """Stands in for the `Costmap2d(self)` subscription (py:118)."""
        self._solver.set_costmap(cells, resolution, origin_x, origin_y)

    def footprint_callback(self, msg):                       # py:154-155
        self.footprint = msg.polygon if hasattr(msg, "polygon") else msg

    def cb_params(self, data):                               # py:405-439
        changes = {}
        for parameter in data:
            # py:407: Parameter.Type.DOUBLE only.  rclpy's Parameter.Type is a plain Enum (DOUBLE.value == 3),
            # stand-ins may carry the int itself
            kind = getattr(parameter, "type_", 3)
            if getattr(kind, "value", kind) != 3:
                continue
            if parameter.name in DYNAMIC_PARAMS:
                changes[parameter.name] = float(parameter.value)
        for k, v in changes.items():                    # the node's attributes always change ...
            setattr(self, k, v)
        if self.reference_quirks:                       # ... the optimisation only sees these
            changes = {k: v for k, v in changes.items() if k in EFFECTIVE_DYNAMIC_PARAMS}
        if changes:
            self._params.update(changes)
            self._solver.set_params(**changes)
        return types.SimpleNamespace(successful=True)

    # -- the service callback -----------------------------------------------------------
    def optimizer(self, request, response=None):             # py:349-403
        if response is None:
            response = make_response()
        current_time = self._clock()                         # py:369-371
        delta_t = current_time - self.last_time
        self.last_time = current_time
        self.control_interval = request.control_interval     # py:355
        rec = request_record(request, delta_t=delta_t)
        fp = None
        pts = getattr(self.footprint, "points", None) if self.footprint is not None else None
        if pts:
            fp = np.array([[[q.x, q.y] for q in pts]], dtype=np.float64)
        cmds, x, path = self._solver.solve(rec, self._state, self._warm, want_path=True, footprints=fp)
        self.last_result = types.SimpleNamespace(x=x[0], fun=float(cmds["cost"][0]),
                                                 success=bool(cmds["status"][0] == 0),
                                                 nit=int(cmds["iterations"][0]), status=int(cmds["status"][0]))
        self.local_plan = path[0]                            # py:293-306 rollout, for `local_plan`
        tw = response.output_vel.twist
        tw.linear.x, tw.linear.y, tw.angular.z = (float(v) for v in cmds["vel"][0])   # py:375-377, 389-391
        return response

    def close(self):
        self._solver.close()
