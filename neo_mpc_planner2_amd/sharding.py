"""Multi-GPU sharding of a batch of independent MPC instances.

Each instance is independent, so the batch is block-partitioned across ranks with no
data-path collective; the costmap and the parameters are replicated.  The only exchange
is one all-gather of the (vx, vy, omega) commands (RCCL over xGMI when the backend is
"nccl"; gloo on CPU in the tests): 24 bytes per instance, e.g. 6.3 MB per rank at
262 144 instances, one direct transfer per peer link.
"""
import numpy as np


def partition(count, world_size, rank):
    """Contiguous block partition: instance b -> rank b // ceil(count / world_size).
    Returns (start, stop)."""
    per = (count + world_size - 1) // world_size
    start = min(count, rank * per)
    return start, min(count, start + per)


def gather_commands(local, out=None, group=None, async_op=False):
    """All-gather the local (n_local, 3) command tensor of every rank into
    out[world, n_local, 3] (allocated when None).  Equal shard sizes are required.
    With `async_op` the collective runs on the backend's own stream and the work handle is
    returned as well, so the next tick's solve kernel overlaps this tick's gather."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    local = local.contiguous()
    if out is None:
        out = torch.empty((world,) + tuple(local.shape), dtype=local.dtype, device=local.device)
    if local.is_cuda and dist.get_backend(group) == "gloo":
        # gloo gathers host tensors only: staged through the host.  (What `bench.py --gpus N` uses when every rank shares
        # ONE device -- NEO_MPC_BENCH_SHARE_DEVICE=1, RCCL refuses that --, so that the world > 1 path runs on a 1-GPU box.)
        host = torch.empty((world,) + tuple(local.shape), dtype=local.dtype)
        dist.all_gather_into_tensor(host.view((-1,) + tuple(local.shape[1:])), local.cpu(), group=group)
        out.copy_(host)
        return (out, _Done()) if async_op else out
    # concatenated layout [world * n_local, 3] (the form both RCCL and gloo accept)
    work = dist.all_gather_into_tensor(out.view((-1,) + tuple(local.shape[1:])), local, group=group,
                                       async_op=async_op)
    return (out, work) if async_op else out


class _Done:
    """work handle of a collective that has already completed"""

    def wait(self):
        return True


def solve_sharded(solve_fn, problems, states, warm, group=None):
    """Solve a host batch that every rank holds in full: each rank solves its block with
    `solve_fn(problems, states, warm) -> (commands, solution)` and the velocity commands of
    all instances are returned on every rank (one all-gather).  `states` / `warm` are updated
    in place for the local block only (per-instance state stays resident on its rank)."""
    import torch
    import torch.distributed as dist
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    count = len(problems)
    per = (count + world - 1) // world
    lo, hi = partition(count, world, rank)
    local = np.zeros((per, 3), dtype=np.float64)
    if hi > lo:
        st, wm = states[lo:hi].copy(), warm[lo:hi].copy()
        cmds, _ = solve_fn(problems[lo:hi], st, wm)
        states[lo:hi] = st
        warm[lo:hi] = wm
        local[: hi - lo] = cmds["vel"]
    out = gather_commands(torch.from_numpy(local), group=group)
    return out.numpy().reshape(world * per, 3)[:count]
