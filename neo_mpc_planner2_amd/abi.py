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
    ("reserved", "<f8", (7,)),
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
    ("reserved_i", "<i4"),
    ("reserved", "<f8", (3,)),
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

STATUS_CONVERGED = 0
STATUS_MAX_ITER = 1

FLAG_RESET = 1
FLAG_STOPPED = 2


def new_states(count, control_steps, waiting_time=3.0):
    """Fresh node state for `count` instances (py:115-152) and their warm starts (py:136)."""
    st = np.zeros(count, dtype=STATE_DTYPE)
    st["waiting_time"] = waiting_time
    warm = np.zeros((count, 3 * control_steps), dtype=np.float64)
    return st, warm
