"""Batched MPC solver handle: thin Python host side over the C-ABI (include/neo_mpc.h).

`BatchSolver` is what bench.py, the tests and `MpcOptimizationServer` (the mirror of
the reference's service node) drive.  Host (NumPy) batches go through
`neo_mpc_solve_batch`; device-resident batches (torch CUDA tensors, plumbing only)
through `neo_mpc_solve_batch_device` on torch's current stream.
"""
import ctypes as C

import numpy as np

from . import _lib, abi


class BatchSolver:
    """One solver configuration + one costmap on one GPU (a `neo_mpc_handle`)."""

    def __init__(self, params=None, device=0, **overrides):
        self._lib = _lib.load()
        self._abi = int(self._lib.neo_mpc_abi_version())
        self._handle = None
        self.params = dict(params or {})
        self.params.update(overrides)
        ps = abi.params_struct(self.params)
        self.control_steps = int(ps.control_steps)
        self.device = int(device)
        h = self._lib.neo_mpc_create(C.byref(ps), self.device)
        if not h:
            code = self._lib.neo_mpc_last_error_code() if hasattr(self._lib, "neo_mpc_last_error_code") else -1
            raise _lib.NeoMpcError(code, (self._lib.neo_mpc_last_error() or b"").decode())
        self._handle = C.c_void_p(h)
        self._keep = None
        self._last_call = None
        self._in_flight = {}    # ticket -> (batch struct, arrays): kept alive until solve_wait

    # -- lifecycle ------------------------------------------------------------------
    def close(self):
        if self._handle is not None:
            self._lib.neo_mpc_destroy(self._handle)
            self._handle = None
        # (the marshalled batches keep the caller's arrays alive: let go of them with the handle)
        self._last_call = None
        self._keep = None
        self._in_flight = {}

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # -- configuration --------------------------------------------------------------
    def set_params(self, **changes):
        """Dynamic reconfigure (reference: cb_params, mpc_optimization_server.py:405-439)."""
        new = dict(self.params)
        new.update(changes)
        ps = abi.params_struct(new)
        if int(ps.control_steps) != self.control_steps:
            raise ValueError("control_steps cannot change on a live handle (the reference bakes it at init, py:125-137)")
        _lib.check(self._lib.neo_mpc_set_params(self._handle, C.byref(ps)))
        self.params = new    # (a refused set keeps the old parameters, like the library does)

    def set_costmap(self, cells, resolution, origin_x, origin_y):
        """cells: uint8 [size_y, size_x] raw nav2 costs (NumPy) or a CUDA uint8 torch tensor."""
        if isinstance(cells, np.ndarray):
            cells = np.ascontiguousarray(cells, dtype=np.uint8)
            sy, sx = cells.shape
            _lib.check(self._lib.neo_mpc_set_costmap(self._handle, C.c_void_p(cells.ctypes.data), sx, sy,
                                                    float(resolution), float(origin_x), float(origin_y)))
        else:
            import torch
            assert cells.is_cuda and cells.dtype == torch.uint8 and cells.is_contiguous()
            sy, sx = cells.shape
            stream = torch.cuda.current_stream(cells.device).cuda_stream
            _lib.check(self._lib.neo_mpc_set_costmap_device(
                self._handle, C.c_void_p(cells.data_ptr()), sx, sy, float(resolution), float(origin_x),
                float(origin_y), C.c_void_p(stream)))

    def set_costmap_pool(self, cells, resolution, origins):
        """Fleet variant: cells uint8 [count, size_y, size_x] (NumPy or CUDA torch tensor), origins
        float64 [count, 2]; instances pick their map with `problems["map_index"]`.  With device
        tensors `origins` is read by every later solve and must stay alive (it may be updated in
        place between ticks, as rolling windows move)."""
        if isinstance(cells, np.ndarray):
            cells = np.ascontiguousarray(cells, dtype=np.uint8)
            origins = np.ascontiguousarray(origins, dtype=np.float64)
            m, sy, sx = cells.shape
            assert origins.shape == (m, 2)
            _lib.check(self._lib.neo_mpc_set_costmap_pool(self._handle, C.c_void_p(cells.ctypes.data), m, sx, sy,
                                                         float(resolution), C.c_void_p(origins.ctypes.data)))
        else:
            import torch
            assert cells.is_cuda and cells.dtype == torch.uint8 and cells.is_contiguous()
            assert origins.is_cuda and origins.dtype == torch.float64 and origins.is_contiguous()
            m, sy, sx = cells.shape
            assert tuple(origins.shape) == (m, 2)
            self._pool_origins = origins   # keep it alive
            stream = torch.cuda.current_stream(cells.device).cuda_stream
            _lib.check(self._lib.neo_mpc_set_costmap_pool_device(
                self._handle, C.c_void_p(cells.data_ptr()), m, sx, sy, float(resolution),
                C.c_void_p(origins.data_ptr()), C.c_void_p(stream)))

    HOST_PATHS = {"auto": 0, "staged": 1, "zerocopy": 2, "zerocopy_out": 3}

    def set_host_path(self, mode):
        """How `solve` moves a batch whose arrays are all page-locked: "auto" (= "zerocopy": K1 works on the
        caller's arrays in place over PCIe), "staged" (DMA copies in and out), "zerocopy_out" (DMA in, results
        written in place)."""
        _lib.check(self._lib.neo_mpc_set_host_path(self._handle, self.HOST_PATHS[mode]))

    def kernel_info(self):
        a, b, c = C.c_uint32(), C.c_uint32(), C.c_uint32()
        _lib.check(self._lib.neo_mpc_kernel_info(self._handle, C.byref(a), C.byref(b), C.byref(c)))
        return dict(lds_bytes=a.value, reach_cells=b.value, tile_in_lds=bool(c.value))

    # -- host batches ----------------------------------------------------------------
    def _host_batch(self, problems, states, warm, solution, want_path, footprints, commands=None):
        n = self.control_steps
        problems = np.ascontiguousarray(problems, dtype=abi.PROBLEM_DTYPE)
        count = problems.shape[0]
        assert states.dtype == abi.STATE_DTYPE and states.shape == (count,) and states.flags.c_contiguous
        assert warm.dtype == np.float64 and warm.shape == (count, 3 * n) and warm.flags.c_contiguous
        if commands is None:
            commands = np.zeros(count, dtype=abi.COMMAND_DTYPE)
        assert commands.dtype == abi.COMMAND_DTYPE and commands.shape == (count,) and commands.flags.c_contiguous
        path = np.zeros((count, n, 3)) if want_path else None
        if footprints is not None:
            footprints = np.ascontiguousarray(footprints, dtype=np.float64)
            assert footprints.shape[0] == count and footprints.shape[2] == 2
        b = abi.batch_struct(problems, states, warm, commands, solution, path, footprints)
        self._keep = (problems, footprints)
        return b, commands, path

    def solve(self, problems, states, warm, want_path=False, footprints=None, out=None):
        """`optimizer()` (py:349-403) for a batch of host records.  `states` / `warm` are
        updated in place.  Returns (commands, solution[, path]).  `out` = (commands, solution) arrays to fill
        instead of fresh ones -- page-locked ones (with page-locked inputs) make every transfer of the call a DMA."""
        count = len(problems)
        if out is not None and not want_path and footprints is None:
            # a caller that owns its arrays (a fleet server's request arena) passes the same ones every tick: the
            # marshalled batch of the previous call is reused as long as every array is the same object
            # (address and shape are part of the key: an array resized or re-allocated in place keeps its id())
            key = tuple((id(a), a.ctypes.data, a.shape) for a in (problems, states, warm, out[0], out[1]))
            if self._last_call is not None and self._last_call[0] == key:
                _lib.check(self._lib.neo_mpc_solve_batch(self._handle, self._last_call[1]))
                return out
        commands, solution = out if out is not None else (None, np.zeros((count, 3 * self.control_steps)))
        assert solution.dtype == np.float64 and solution.shape == (count, 3 * self.control_steps) and solution.flags.c_contiguous
        b, commands, path = self._host_batch(problems, states, warm, solution, want_path, footprints, commands)
        if out is not None and not want_path and footprints is None and self._keep[0] is problems:
            self._last_call = (key, C.byref(b), b, (problems, states, warm, out))   # (keeps the arrays and the struct alive)
        _lib.check(self._lib.neo_mpc_solve_batch(self._handle, C.byref(b)))
        return (commands, solution, path) if want_path else (commands, solution)

    def solve_begin(self, problems, states, warm, out):
        """First half of `solve` for page-locked arrays (`client->async_send_request(request)`, cpp:248): enqueues the
        batch -- it is worked on in place -- and returns a ticket for `solve_wait`.  `out` = (commands, solution).  The
        arrays must stay untouched (and alive) until the wait; up to four batches may be in flight."""
        assert problems.dtype == abi.PROBLEM_DTYPE and problems.flags.c_contiguous
        b, _, _ = self._host_batch(problems, states, warm, out[1], False, None, out[0])
        ticket = C.c_uint32(0)
        _lib.check(self._lib.neo_mpc_solve_batch_begin(self._handle, C.byref(b), C.byref(ticket)))
        self._in_flight[ticket.value] = (b, problems, states, warm, out)
        return ticket.value

    def solve_wait(self, ticket):
        """Second half (`result.get()`, cpp:250): blocks until the batch of `ticket` is done; its results are then in the
        arrays handed to `solve_begin`, which are returned."""
        try:
            _lib.check(self._lib.neo_mpc_solve_batch_wait(self._handle, C.c_uint32(ticket)))
        except Exception:
            self._in_flight.pop(ticket, None)   # (the library has given the slot up either way: no stale entry)
            raise
        return self._in_flight.pop(ticket)[4]

    def postprocess(self, problems, states, warm, solution, success=None, want_path=False, footprints=None):
        """Everything of `optimizer()` after the solve (py:365-403) with `solution` = x.x."""
        solution = np.ascontiguousarray(solution, dtype=np.float64)
        b, commands, path = self._host_batch(problems, states, warm, solution, want_path, footprints)
        sp = None
        if success is not None:
            success = np.ascontiguousarray(success, dtype=np.int32)
            sp = C.c_void_p(success.ctypes.data)
        _lib.check(self._lib.neo_mpc_postprocess_batch(self._handle, C.byref(b), sp))
        return (commands, path) if want_path else commands

    def objective(self, problems, u):
        """`objective()` (py:204-269) on the device for u[count, 3*control_steps]."""
        problems = np.ascontiguousarray(problems, dtype=abi.PROBLEM_DTYPE)
        u = np.ascontiguousarray(u, dtype=np.float64)
        out = np.zeros(len(problems))
        _lib.check(self._lib.neo_mpc_objective_batch(
            self._handle, C.c_void_p(problems.ctypes.data), C.c_void_p(u.ctypes.data),
            C.c_void_p(out.ctypes.data), len(problems)))
        return out

    def gradient(self, problems, u):
        """Test hook: the total gradient the solve kernel works with at `u` (projected like x0):
        analytic adjoint gradient + gradient of the control norm, from inside the kernel variant
        the current parameters select."""
        problems = np.ascontiguousarray(problems, dtype=abi.PROBLEM_DTYPE)
        u = np.ascontiguousarray(u, dtype=np.float64)
        out = np.zeros_like(u)
        _lib.check(self._lib.neo_mpc_gradient_batch(
            self._handle, C.c_void_p(problems.ctypes.data), C.c_void_p(u.ctypes.data),
            C.c_void_p(out.ctypes.data), len(problems)))
        return out

    def direction(self, problems, u, iteration):
        """Test hook: the search direction of lanes 32-63 in solver iteration `iteration` (0-based) of a
        solve started from `u`; rows of instances that stopped earlier stay NaN."""
        problems = np.ascontiguousarray(problems, dtype=abi.PROBLEM_DTYPE)
        u = np.ascontiguousarray(u, dtype=np.float64)
        out = np.full_like(u, np.nan)
        _lib.check(self._lib.neo_mpc_direction_batch(
            self._handle, C.c_void_p(problems.ctypes.data), C.c_void_p(u.ctypes.data),
            C.c_void_p(out.ctypes.data), len(problems), int(iteration)))
        return out

    # -- carrot selection (the step before the solver) ------------------------------------
    def select_carrots(self, plan_poses, plan_offsets, robot_poses, slow_down, footprint_costs=None,
                       problems=None, lookahead_dist_min=0.5, lookahead_dist_max=0.5,
                       lookahead_dist_close_to_goal=0.5, max_transform_dist=1e9):
        """Plan pruning + look-ahead point + slow_down_ update for a ragged batch of plans
        (NeoMpcPlanner.cpp:83-104, 157-189, 221-232).  `slow_down` (int32) is updated in place;
        when `problems` is given the carrot pose is written into the requests."""
        plan_poses = np.ascontiguousarray(plan_poses, dtype=np.float64).reshape(-1, 3)
        plan_offsets = np.ascontiguousarray(plan_offsets, dtype=np.uint32)
        robot_poses = np.ascontiguousarray(robot_poses, dtype=np.float64).reshape(-1, 3)
        count = robot_poses.shape[0]
        assert plan_offsets.shape == (count + 1,) and slow_down.dtype == np.int32 and slow_down.shape == (count,)
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
            assert problems.dtype == abi.PROBLEM_DTYPE and problems.flags.c_contiguous
            b.problems = problems.ctypes.data
        _lib.check(self._lib.neo_mpc_select_carrots(self._handle, C.byref(lp), C.byref(b)))
        return carrots

    def select_carrots_device(self, lp, plan_poses, plan_offsets, robot_poses, slow_down, carrots,
                              footprint_costs=None, problems=None, stream=None):
        """Device-resident variant: torch CUDA tensors; enqueues K4 on `stream`."""
        import torch
        b = abi.NeoMpcPlanBatch()
        b.count = robot_poses.shape[0]
        b.plan_poses = plan_poses.data_ptr()
        b.plan_offsets = plan_offsets.data_ptr()
        b.robot_poses = robot_poses.data_ptr()
        b.footprint_costs = footprint_costs.data_ptr() if footprint_costs is not None else None
        b.slow_down = slow_down.data_ptr()
        b.carrots = carrots.data_ptr()
        b.problems = problems.data_ptr() if problems is not None else None
        if stream is None:
            stream = torch.cuda.current_stream(robot_poses.device).cuda_stream
        _lib.check(self._lib.neo_mpc_select_carrots_device(self._handle, C.byref(lp), C.byref(b), C.c_void_p(stream)))

    # -- device-resident batches (torch tensors as plain device memory) ----------------
    def balance_dispatch(self, commands, stream=None):
        """Dispatch order of the following device solves of the same count from the iteration counts in `commands` (a CUDA
        uint8 tensor of command records, e.g. the previous tick's; None: back to launch order) --
        neo_mpc_balance_dispatch_device.  Enqueued on `stream` (default: torch's current).  Changes no result."""
        if commands is None:
            _lib.check(self._lib.neo_mpc_balance_dispatch_device(self._handle, None, 0, None))
            return
        import torch
        if stream is None:
            stream = torch.cuda.current_stream(commands.device).cuda_stream
        _lib.check(self._lib.neo_mpc_balance_dispatch_device(self._handle, C.c_void_p(commands.data_ptr()), commands.shape[0],
                                                             C.c_void_p(stream)))

    def solve_device(self, problems, states, warm, commands, solution=None, path=None, footprints=None,
                     stream=None, velocities=None, events=None):
        """All arguments are CUDA uint8/float64 torch tensors holding the C records
        (`DeviceBatch` builds them).  Enqueues K1 on `stream` (default: torch's current).
        `events` = (start, stop) torch.cuda.Event pair, already created (recorded once): stamped by
        the kernel dispatch itself (`neo_mpc_solve_batch_device_timed`)."""
        import torch
        count = problems.shape[0]
        b = abi.NeoMpcBatch()
        b.count = count
        b.problems = problems.data_ptr()
        b.states = states.data_ptr()
        b.warm_start = warm.data_ptr()
        b.commands = commands.data_ptr()
        b.solution = solution.data_ptr() if solution is not None else None
        b.predicted_path = path.data_ptr() if path is not None else None
        if footprints is not None:
            b.footprints = footprints.data_ptr()
            b.footprint_points = footprints.shape[1]
        if velocities is not None:
            b.velocities = velocities.data_ptr()
        if stream is None:
            stream = torch.cuda.current_stream(problems.device).cuda_stream
        if events is None:
            _lib.check(self._lib.neo_mpc_solve_batch_device(self._handle, C.byref(b), C.c_void_p(stream)))
        else:
            _lib.check(self._lib.neo_mpc_solve_batch_device_timed(
                self._handle, C.byref(b), C.c_void_p(stream), C.c_void_p(events[0].cuda_event),
                C.c_void_p(events[1].cuda_event)))


class DeviceBatch:
    """A batch resident in HBM: the C records as torch CUDA byte/float64 tensors."""

    def __init__(self, problems, states, warm, device, want_solution=True):
        import torch
        self.count = len(problems)
        dev = torch.device(device)
        self.problems = torch.from_numpy(np.ascontiguousarray(problems).view(np.uint8).reshape(self.count, -1)).to(dev)
        self.states = torch.from_numpy(np.ascontiguousarray(states).view(np.uint8).reshape(self.count, -1)).to(dev)
        self.warm = torch.from_numpy(np.ascontiguousarray(warm)).to(dev)
        self.commands = torch.zeros((self.count, abi.COMMAND_DTYPE.itemsize), dtype=torch.uint8, device=dev)
        self.solution = torch.zeros_like(self.warm) if want_solution else None
        self.vel = torch.zeros((self.count, 3), dtype=torch.float64, device=dev)   # packed (vx, vy, w)

    def fresh_state(self):
        """Another (states, warm, commands) set for the same problems (shares `problems`)."""
        import torch
        other = object.__new__(DeviceBatch)
        other.count = self.count
        other.problems = self.problems
        other.states = self.states.clone()
        other.warm = self.warm.clone()
        other.commands = torch.zeros_like(self.commands)
        other.solution = None
        other.vel = torch.zeros_like(self.vel)
        return other

    def commands_host(self):
        return self.commands.cpu().numpy().view(abi.COMMAND_DTYPE).reshape(self.count)

    def velocities(self):
        """(count, 3) float64 view of the (vx, vy, omega) outputs on the device."""
        import torch
        return self.commands.view(torch.float64).view(self.count, 6)[:, :3]

    def states_host(self):
        return self.states.cpu().numpy().view(abi.STATE_DTYPE).reshape(self.count)
