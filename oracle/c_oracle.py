"""ctypes front end of the plain-C oracle (oracle/mpc_oracle.c).

TEST INFRASTRUCTURE ONLY (see the header of mpc_oracle.c).  `load()` builds
oracle/_build/libmpc_oracle.so with gcc when it is missing or stale.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from neo_mpc_planner2_amd import abi

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "_build", "libmpc_oracle.so")
_lib = None


def build():
    src = os.path.join(HERE, "mpc_oracle.c")
    hdrs = [os.path.join(HERE, "..", "include", "neo_mpc.h"),
            os.path.join(HERE, "..", "neo_mpc_planner2_amd", "csrc", "solver_rules.h")]   # (the rule book the mirror shares)
    if (not os.path.exists(LIB)) or os.path.getmtime(LIB) < max(os.path.getmtime(f) for f in [src] + hdrs):
        subprocess.check_call(["make", "-C", HERE, "-s"])
    return LIB


def load():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        _lib.orc_yaw.restype = C.c_double
        _lib.orc_yaw.argtypes = [C.c_double] * 4
    return _lib


def _map_args(cmap):
    cells, res, ox, oy = cmap
    cells = np.ascontiguousarray(cells, dtype=np.uint8)
    return cells, [cells.ctypes.data_as(C.c_void_p), C.c_int32(cells.shape[1]), C.c_int32(cells.shape[0]),
                   C.c_double(res), C.c_double(ox), C.c_double(oy)]


def set_threads(n):
    """OpenMP threads of solve_batch (bench.py: the CPUs this process may use, not the host's thread count)."""
    load().orc_set_threads(C.c_int(int(n)))


def yaw(x, y, z, w):
    return load().orc_yaw(x, y, z, w)


def objective_batch(params, cmap, problems, u):
    lib = load()
    ps = abi.params_struct(params)
    cells, margs = _map_args(cmap)
    problems = np.ascontiguousarray(problems)
    u = np.ascontiguousarray(u, dtype=np.float64)
    out = np.zeros(len(problems))
    lib.orc_objective_batch(C.byref(ps), *margs, C.c_void_p(problems.ctypes.data),
                            C.c_void_p(u.ctypes.data), C.c_void_p(out.ctypes.data),
                            C.c_size_t(len(problems)))
    return out


def route_batch(params, cmap, problems):
    """1 where AUTO at control_steps 3 sends the instance to the stage-wise direction (its reach tile is not all free), 0
    where it takes the dense one (solver_rules.h neo_rules_routes_by_neighbourhood)."""
    lib = load()
    ps = abi.params_struct(params)
    cells, margs = _map_args(cmap)
    problems = np.ascontiguousarray(problems)
    out = np.zeros(len(problems), dtype=np.int32)
    lib.orc_route_batch(C.byref(ps), *margs, C.c_void_p(problems.ctypes.data), C.c_void_p(out.ctypes.data),
                        C.c_size_t(len(problems)))
    return out


def footprint_cost_batch(cmap, pts):
    lib = load()
    cells, margs = _map_args(cmap)
    pts = np.ascontiguousarray(pts, dtype=np.float64)
    out = np.zeros(pts.shape[0])
    lib.orc_footprint_cost_batch(*margs, C.c_void_p(pts.ctypes.data), C.c_int(pts.shape[1]),
                                 C.c_void_p(out.ctypes.data), C.c_size_t(pts.shape[0]))
    return out


def _run(fn, params, cmap, problems, states, warm, solution=None, want_path=False,
         footprints=None, extra=()):
    ps = abi.params_struct(params)
    n = ps.control_steps
    cells, margs = _map_args(cmap)
    problems = np.ascontiguousarray(problems)
    count = len(problems)
    commands = np.zeros(count, dtype=abi.COMMAND_DTYPE)
    if solution is None:
        solution = np.zeros((count, 3 * n))
    path = np.zeros((count, n, 3)) if want_path else None
    if footprints is not None:
        footprints = np.ascontiguousarray(footprints, dtype=np.float64)
    b = abi.batch_struct(problems, states, warm, commands, solution, path, footprints)
    fn(C.byref(ps), *margs, C.byref(b), *extra)
    return commands, solution, path


def solve_batch(params, cmap, problems, states, warm, want_path=False, footprints=None):
    """Full optimizer() path with the build's solver (CPU mirror).  states/warm updated in place."""
    return _run(load().orc_solve_batch, params, cmap, problems, states, warm,
                want_path=want_path, footprints=footprints)


def postprocess_batch(params, cmap, problems, states, warm, solution, success=None,
                      want_path=False, footprints=None):
    """optimizer() after the solve (py:365-403) with injected solver output."""
    solution = np.ascontiguousarray(solution, dtype=np.float64)
    if success is None:
        sp = C.c_void_p(None)
        keep = None
    else:
        keep = np.ascontiguousarray(success, dtype=np.int32)
        sp = C.c_void_p(keep.ctypes.data)
    return _run(load().orc_postprocess_batch, params, cmap, problems, states, warm, solution=solution,
                want_path=want_path, footprints=footprints, extra=(sp,))


def select_carrots(plan_poses, plan_offsets, robot_poses, slow_down, footprint_costs=None, problems=None,
                   lookahead_dist_min=0.5, lookahead_dist_max=0.5, lookahead_dist_close_to_goal=0.5,
                   max_transform_dist=1e9):
    """cpp:83-104, 157-189, 221-232 restated (oracle/mpc_oracle.c part 3); slow_down updated in place."""
    lib = load()
    plan_poses = np.ascontiguousarray(plan_poses, dtype=np.float64).reshape(-1, 3)
    plan_offsets = np.ascontiguousarray(plan_offsets, dtype=np.uint32)
    robot_poses = np.ascontiguousarray(robot_poses, dtype=np.float64).reshape(-1, 3)
    count = robot_poses.shape[0]
    carrots = np.zeros(count, dtype=abi.CARROT_DTYPE)
    lp = abi.NeoMpcLookaheadParams(lookahead_dist_min, lookahead_dist_max, lookahead_dist_close_to_goal,
                                   max_transform_dist)
    b = abi.NeoMpcPlanBatch()
    b.count = count
    b.plan_poses = plan_poses.ctypes.data
    b.plan_offsets = plan_offsets.ctypes.data
    b.robot_poses = robot_poses.ctypes.data
    if footprint_costs is not None:
        footprint_costs = np.ascontiguousarray(footprint_costs, dtype=np.float64)
        b.footprint_costs = footprint_costs.ctypes.data
    b.slow_down = slow_down.ctypes.data
    b.carrots = carrots.ctypes.data
    if problems is not None:
        b.problems = problems.ctypes.data
    lib.orc_select_carrots(C.byref(lp), C.byref(b))
    return carrots
