"""ctypes binding of the C-ABI library `libneo_mpc.so` (include/neo_mpc.h).

The library is built in-tree by `__graft_entry__.build()` /
`make -C neo_mpc_planner2_amd/csrc`.  There is no Python or CPU fallback: when the
shared object is missing, `load()` raises.
"""
import ctypes as C
import os

from . import abi

HERE = os.path.dirname(os.path.abspath(__file__))
#: NEO_MPC_LIB selects another build of the same library (development A/B runs inside one gpurun
#: call: box-to-box variance is ~5 %, so two builds are only comparable on the same box)
LIB_PATH = os.environ.get("NEO_MPC_LIB") or os.path.join(HERE, "libneo_mpc.so")

#: every symbol include/neo_mpc.h declares
EXPORTS = (
    "neo_mpc_abi_version", "neo_mpc_behaviour_version", "neo_mpc_effective_method", "neo_mpc_balance_dispatch_device", "neo_mpc_last_error", "neo_mpc_last_error_code", "neo_mpc_default_params", "neo_mpc_create",
    "neo_mpc_destroy", "neo_mpc_set_params", "neo_mpc_get_params", "neo_mpc_set_costmap",
    "neo_mpc_set_costmap_device", "neo_mpc_set_costmap_pool", "neo_mpc_set_costmap_pool_device", "neo_mpc_solve_batch", "neo_mpc_solve_batch_device",
    "neo_mpc_solve_batch_device_timed", "neo_mpc_solve_batch_begin", "neo_mpc_solve_batch_wait",
    "neo_mpc_postprocess_batch", "neo_mpc_objective_batch", "neo_mpc_gradient_batch", "neo_mpc_direction_batch",
    "neo_mpc_kernel_info", "neo_mpc_pin_host_memory", "neo_mpc_unpin_host_memory", "neo_mpc_set_host_path",
    "neo_mpc_select_carrots", "neo_mpc_select_carrots_device",
    "neo_mpc_rccl_available", "neo_mpc_comm_init_all", "neo_mpc_comm_destroy", "neo_mpc_group_start",
    "neo_mpc_group_end", "neo_mpc_allgather_velocities", "neo_mpc_broadcast_costmap",
)

_lib = None


class NeoMpcError(RuntimeError):
    def __init__(self, code, message):
        super().__init__("neo_mpc error %d: %s" % (code, message))
        self.code = code


def load():
    """Load libneo_mpc.so and declare the prototypes.  Raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "%s is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C neo_mpc_planner2_amd/csrc`).  There is no CPU fallback." % LIB_PATH)
    # PyTorch wheels bundle their own HIP runtime; if torch is going to be used in this process
    # (device memory / streams / torch.distributed plumbing) it must initialise that runtime
    # BEFORE another copy is mapped, otherwise torch reports "No HIP GPUs are available".
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(LIB_PATH)
    P = C.POINTER
    lib.neo_mpc_abi_version.restype = C.c_int
    lib.neo_mpc_last_error.restype = C.c_char_p
    lib.neo_mpc_default_params.argtypes = [P(abi.NeoMpcParams)]
    lib.neo_mpc_create.restype = C.c_void_p
    lib.neo_mpc_create.argtypes = [P(abi.NeoMpcParams), C.c_int]
    lib.neo_mpc_destroy.restype = None
    lib.neo_mpc_destroy.argtypes = [C.c_void_p]
    lib.neo_mpc_set_params.argtypes = [C.c_void_p, P(abi.NeoMpcParams)]
    lib.neo_mpc_get_params.argtypes = [C.c_void_p, P(abi.NeoMpcParams)]
    lib.neo_mpc_set_costmap.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32,
                                        C.c_double, C.c_double, C.c_double]
    lib.neo_mpc_set_costmap_device.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32,
                                               C.c_double, C.c_double, C.c_double, C.c_void_p]
    lib.neo_mpc_set_costmap_pool.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32,
                                             C.c_double, C.c_void_p]
    lib.neo_mpc_set_costmap_pool_device.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32,
                                                    C.c_double, C.c_void_p, C.c_void_p]
    lib.neo_mpc_solve_batch.argtypes = [C.c_void_p, P(abi.NeoMpcBatch)]
    lib.neo_mpc_solve_batch_device.argtypes = [C.c_void_p, P(abi.NeoMpcBatch), C.c_void_p]
    lib.neo_mpc_solve_batch_device_timed.argtypes = [C.c_void_p, P(abi.NeoMpcBatch), C.c_void_p, C.c_void_p, C.c_void_p]
    lib.neo_mpc_postprocess_batch.argtypes = [C.c_void_p, P(abi.NeoMpcBatch), C.c_void_p]
    lib.neo_mpc_objective_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
    lib.neo_mpc_gradient_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
    lib.neo_mpc_direction_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    lib.neo_mpc_select_carrots.argtypes = [C.c_void_p, P(abi.NeoMpcLookaheadParams), P(abi.NeoMpcPlanBatch)]
    lib.neo_mpc_select_carrots_device.argtypes = [C.c_void_p, P(abi.NeoMpcLookaheadParams),
                                                  P(abi.NeoMpcPlanBatch), C.c_void_p]
    lib.neo_mpc_solve_batch_begin.argtypes = [C.c_void_p, P(abi.NeoMpcBatch), P(C.c_uint32)]
    lib.neo_mpc_solve_batch_wait.argtypes = [C.c_void_p, C.c_uint32]
    lib.neo_mpc_pin_host_memory.argtypes = [C.c_void_p, C.c_size_t]
    lib.neo_mpc_unpin_host_memory.argtypes = [C.c_void_p]
    lib.neo_mpc_set_host_path.argtypes = [C.c_void_p, C.c_int]
    lib.neo_mpc_kernel_info.argtypes = [C.c_void_p, P(C.c_uint32), P(C.c_uint32), P(C.c_uint32)]
    if os.environ.get("NEO_MPC_LIB"):
        # development A/B against an older build of the same library (tools/ab_many.sh): the record layouts have not
        # changed since ABI 1, so only the entry points a tool actually calls have to exist
        if lib.neo_mpc_abi_version() not in (1, abi.ABI_VERSION):
            raise ImportError("%s: ABI version %d" % (LIB_PATH, lib.neo_mpc_abi_version()))
    else:
        for name in EXPORTS:
            getattr(lib, name)     # AttributeError here = the .so is older than include/neo_mpc.h
        if lib.neo_mpc_abi_version() != abi.ABI_VERSION:
            raise ImportError("libneo_mpc.so ABI version %d, this package binds %d" % (lib.neo_mpc_abi_version(), abi.ABI_VERSION))
        lib.neo_mpc_behaviour_version.restype = C.c_int
        lib.neo_mpc_effective_method.restype = C.c_int
        lib.neo_mpc_effective_method.argtypes = [C.c_void_p]
        lib.neo_mpc_balance_dispatch_device.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    _lib = lib
    return lib


def check(code):
    if code != 0:
        raise NeoMpcError(code, (load().neo_mpc_last_error() or b"").decode())
