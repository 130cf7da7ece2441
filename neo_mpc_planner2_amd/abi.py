"""Binary layout shared by the Python host side and the C-ABI (`include/neo_mpc.h`).

The records mirror the fields of the reference's ``neo_srvs2/srv/Optimizer`` request
and response (call sites src/NeoMpcPlanner.cpp:240-252 and
neo_mpc_planner2/mpc_optimization_server.py:349-403) plus the per-node state the
reference keeps between calls (mpc_optimization_server.py:115-152).
All floating point fields are float64, like the ROS messages.
"""
import numpy as np

#: one Optimizer.srv request -> `neo_mpc_problem` (256 bytes)
PROBLEM_DTYPE = np.dtype([
    ("cur_xy", "<f8", (2,)),        # request.current_pose.pose.position.{x,y}   (costmap frame)
    ("cur_q", "<f8", (4,)),         # request.current_pose.pose.orientation x,y,z,w
    ("carrot_xy", "<f8", (2,)),     # request.carrot_pose.pose.position.{x,y}    (base frame)
    ("carrot_q", "<f8", (4,)),      # request.carrot_pose.pose.orientation x,y,z,w
    ("goal_xyz", "<f8", (3,)),      # request.goal_pose.position x,y,z           (plan frame)
    ("goal_q", "<f8", (4,)),        # request.goal_pose.orientation x,y,z,w
    ("cur_vel", "<f8", (3,)),       # request.current_vel linear.x, linear.y, angular.z
    ("control_interval", "<f8"),    # request.control_interval = 1/controller_frequency
    ("delta_t", "<f8"),             # wall-clock seconds since the previous call (py:369-371)
    ("footprint_cost", "<f8"),      # getFootprintCost(published footprint), normalised; used when
                                    # no polygon is supplied (py:262, 343)
    ("map_index", "<i4"),           # which costmap of a pool (BatchSolver.set_costmap_pool); else ignored
    ("switch_opt", "<i4"),          # request.switch_opt (cpp:245; stored py:354, never read)
    ("skip", "<i4"),                # 1: no request is made for this robot this tick (cpp:234-236); state untouched.  0: solve.
                                    # Anything else: host batches are refused, device batches solve the robot (include/neo_mpc.h)
    ("reserved_i", "<i4"),
    ("reserved", "<f8", (5,)),
], align=False)
assert PROBLEM_DTYPE.itemsize == 256

#: per-instance persistent node state -> `neo_mpc_state` (128 bytes); the warm start
#: (py:136 `initial_guess`) is a separate float64[3*control_steps] row per instance.
STATE_DTYPE = np.dtype([
    ("last_control", "<f8", (3,)),  # py:117
    ("old_goal", "<f8", (7,)),      # py:146 / py:402: goal position xyz + orientation xyzw
    ("waiting_time", "<f8"),        # py:103, 361, 378-382
    ("has_old_goal", "<i4"),        # 0 until the first call (py:146 PoseStamped != Pose)
    ("collision", "<i4"),           # py:148 latch
    ("collision_footprint", "<i4"), # py:149
    ("has_prev_u0", "<i4"),           # the build's own hint (no node attribute): 1 = prev_u0 is there
    ("prev_u0", "<f8", (3,)),         # ... the previous solve's first control block before the low-pass (py:366-367)
], align=False)
assert STATE_DTYPE.itemsize == 128

#: Optimizer.srv response + solver diagnostics -> `neo_mpc_command` (48 bytes)
COMMAND_DTYPE = np.dtype([
    ("vel", "<f8", (3,)),           # response.output_vel.twist linear.x, linear.y, angular.z
    ("cost", "<f8"),                # objective value at the raw solver output (`x.fun`)
    ("status", "<i4"),              # NEO_MPC_STATUS_* (0 converged -> `x.success`)
    ("iterations", "<i4"),          # projected-gradient iterations (`x.nit`)
    ("evaluations", "<i4"),         # objective evaluations per lane (`x.nfev` analogue)
    ("flags", "<i4"),               # bit0 reset-on-new-goal taken, bit1 stopped by collision latch
], align=False)
assert COMMAND_DTYPE.itemsize == 48

#: `neo_mpc_carrot` (80 bytes): result of the plan pruning + look-ahead selection
#: (NeoMpcPlanner.cpp:83-104, 157-189, 221-232)
CARROT_DTYPE = np.dtype([
    ("xy", "<f8", (2,)), ("q", "<f8", (4,)), ("lookahead_dist", "<f8"),
    ("begin", "<u4"), ("end", "<u4"), ("closer_to_goal", "<i4"), ("slow_down", "<i4"),
    ("status", "<i4"), ("reserved", "<i4"),
], align=False)
assert CARROT_DTYPE.itemsize == 80

STATUS_CONVERGED = 0
STATUS_MAX_ITER = 1

FLAG_RESET = 1
FLAG_STOPPED = 2
FLAG_SKIPPED = 4
FLAG_WALL_IN_REACH = 8   # a lethal cell within the robot's reach tile (AUTO at control_steps 3: solved by the stage-wise direction)

ABI_VERSION = 2          # NEO_MPC_ABI_VERSION of include/neo_mpc.h
COMPAT_ODOM_YAW_GOAL_W = 1
COMPAT_REFERENCE_START = 2


def new_states(count, control_steps, waiting_time=3.0):
    """Fresh node state for `count` instances (py:115-152) and their warm starts (py:136)."""
    st = np.zeros(count, dtype=STATE_DTYPE)
    st["waiting_time"] = waiting_time
    warm = np.zeros((count, 3 * control_steps), dtype=np.float64)
    return st, warm


# ---------------------------------------------------------------------- ctypes mirrors
import ctypes as _C  # noqa: E402


class NeoMpcParams(_C.Structure):
    """`neo_mpc_params` (include/neo_mpc.h): the reference node's ROS parameters, same
    names (mpc_optimization_server.py:49-75), then the build's solver options."""
    _fields_ = [(n, _C.c_double) for n in (
        "acc_x_limit", "acc_y_limit", "acc_theta_limit",
        "min_vel_x", "min_vel_y", "min_vel_trans", "min_vel_theta",
        "max_vel_x", "max_vel_y", "max_vel_trans", "max_vel_theta",
        "w_trans", "w_orient", "w_control", "w_terminal", "w_costmap", "w_footprint",
        "waiting_time", "low_pass_gain", "opt_tolerance", "prediction_horizon")] + [
        ("control_steps", _C.c_int32), ("max_iterations", _C.c_int32),
        ("lbfgs_memory", _C.c_int32), ("compat_flags", _C.c_int32),
        ("step_tolerance", _C.c_double), ("cost_tolerance", _C.c_double),
        ("kink_radius", _C.c_double), ("stall_step", _C.c_double),
        ("method", _C.c_int32), ("reserved_i", _C.c_int32),
        ("window_tolerance", _C.c_double)]


ROS_PARAM_NAMES = tuple(n for n, _ in NeoMpcParams._fields_[:22])
COMPAT_ODOM_YAW_GOAL_W = 1


class NeoMpcBatch(_C.Structure):
    """`neo_mpc_batch` (include/neo_mpc.h)."""
    _fields_ = [("count", _C.c_size_t), ("problems", _C.c_void_p), ("states", _C.c_void_p),
                ("warm_start", _C.c_void_p), ("commands", _C.c_void_p), ("solution", _C.c_void_p),
                ("predicted_path", _C.c_void_p), ("footprints", _C.c_void_p),
                ("footprint_points", _C.c_uint32), ("reserved", _C.c_uint32), ("velocities", _C.c_void_p)]


class NeoMpcLookaheadParams(_C.Structure):
    """`neo_mpc_lookahead_params`: `<plugin>.lookahead_dist_*` (NeoMpcPlanner.cpp:311-322)."""
    _fields_ = [("lookahead_dist_min", _C.c_double), ("lookahead_dist_max", _C.c_double),
                ("lookahead_dist_close_to_goal", _C.c_double), ("max_transform_dist", _C.c_double)]


class NeoMpcPlanBatch(_C.Structure):
    """`neo_mpc_plan_batch` (include/neo_mpc.h)."""
    _fields_ = [("count", _C.c_size_t), ("plan_poses", _C.c_void_p), ("plan_offsets", _C.c_void_p),
                ("robot_poses", _C.c_void_p), ("footprint_costs", _C.c_void_p), ("slow_down", _C.c_void_p),
                ("carrots", _C.c_void_p), ("problems", _C.c_void_p)]


def params_struct(params=None, **over):
    """dict of ROS parameter names (+ solver options) -> NeoMpcParams.  Missing names take
    the reference node's declared defaults (py:49-75)."""
    d = dict(
        acc_x_limit=0.5, acc_y_limit=0.5, acc_theta_limit=0.5,
        min_vel_x=-0.5, min_vel_y=-0.5, min_vel_trans=0.5, min_vel_theta=-0.5,
        max_vel_x=0.5, max_vel_y=0.5, max_vel_trans=0.5, max_vel_theta=0.5,
        w_trans=0.5, w_orient=0.5, w_control=0.5, w_terminal=0.5, w_costmap=0.5,
        w_footprint=2000.0, waiting_time=3.0, low_pass_gain=0.5, opt_tolerance=1e-5,
        prediction_horizon=0.5, control_steps=3,
        max_iterations=100, lbfgs_memory=4, compat_flags=COMPAT_ODOM_YAW_GOAL_W,
        step_tolerance=0.0, cost_tolerance=0.0, kink_radius=0.0, stall_step=0.0, method=0, reserved_i=0,
        window_tolerance=0.0)
    if params:
        d.update(params)
    d.update(over)
    s = NeoMpcParams()
    for name, ctype in NeoMpcParams._fields_:
        if name == "reserved":
            continue
        setattr(s, name, int(d[name]) if ctype is _C.c_int32 else float(d[name]))
    return s


def batch_struct(problems, states, warm, commands, solution=None, predicted_path=None,
                 footprints=None, ptr=lambda a: a.ctypes.data):
    """Build a NeoMpcBatch over arrays (NumPy by default; `ptr` extracts the address)."""
    b = NeoMpcBatch()
    b.count = int(problems.shape[0])
    b.problems = ptr(problems)
    b.states = ptr(states)
    b.warm_start = ptr(warm)
    b.commands = ptr(commands)
    b.solution = ptr(solution) if solution is not None else None
    b.predicted_path = ptr(predicted_path) if predicted_path is not None else None
    if footprints is not None:
        b.footprints = ptr(footprints)
        b.footprint_points = int(footprints.shape[1])
    else:
        b.footprints = None
        b.footprint_points = 0
    return b
