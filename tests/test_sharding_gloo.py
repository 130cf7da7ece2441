"""Multi-rank path on CPU: world_size 2, gloo.  The batch is block-partitioned, each rank
solves its block, and one all-gather returns every instance's (vx, vy, omega) on every rank
(neo_mpc_planner2_amd/sharding.py).  The per-rank solver here is the CPU oracle mirror (tests
may use oracle/); on the GPU box bench.py drives the same sharding code over RCCL."""
import os
import socket

import numpy as np
import pytest

from neo_mpc_planner2_amd import sharding, synthetic


def test_partition_covers_batch_exactly():
    for count in (0, 1, 7, 4096, 4097):
        for world in (1, 2, 3, 8):
            blocks = [sharding.partition(count, world, r) for r in range(world)]
            assert blocks[0][0] == 0 and blocks[-1][1] == count
            for (a0, a1), (b0, b1) in zip(blocks, blocks[1:]):
                assert a1 == b0 and a0 <= a1
            sizes = [b - a for a, b in blocks]
            assert max(sizes) - min(s for s in sizes if s or True) <= max(sizes)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, count, out_dir):
    import torch.distributed as dist
    from oracle import c_oracle, mpc_oracle as orc
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    params = orc.make_params()
    cmap = synthetic.make_costmap(200, seed=21)
    probs = synthetic.make_problems(count, 200, seed=22)
    st, warm = synthetic.make_states(probs, 3)

    def solve_fn(p, s, w):
        cmds, x, _ = c_oracle.solve_batch(params, cmap, p, s, w)
        return cmds, x

    vel = sharding.solve_sharded(solve_fn, probs, st, warm)
    lo, hi = sharding.partition(count, world, rank)
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), vel=vel, warm=warm, lo=lo, hi=hi,
             last=st["last_control"])
    dist.destroy_process_group()


@pytest.mark.parametrize("world,count", [(2, 64), (2, 37), (4, 37), (4, 3)])
def test_sharded_solve_equals_unsharded(tmp_path, world, count):
    """world 4 with 37 instances: unequal shards (10, 10, 10, 7); with 3 instances one rank has nothing to solve and
    still takes part in the all-gather."""
    import torch.multiprocessing as mp
    from oracle import c_oracle, mpc_oracle as orc
    mp.spawn(_worker, args=(world, _free_port(), count, str(tmp_path)), nprocs=world, join=True)
    params = orc.make_params()
    cmap = synthetic.make_costmap(200, seed=21)
    probs = synthetic.make_problems(count, 200, seed=22)
    st, warm = synthetic.make_states(probs, 3)
    cmds, _, _ = c_oracle.solve_batch(params, cmap, probs, st, warm)
    for rank in range(world):
        r = np.load(os.path.join(str(tmp_path), "rank%d.npz" % rank))
        assert (r["vel"] == cmds["vel"]).all()               # every rank holds all commands
        lo, hi = int(r["lo"]), int(r["hi"])
        assert (r["warm"][lo:hi] == warm[lo:hi]).all()       # per-instance state stays on its rank
        assert (r["last"][lo:hi] == st["last_control"][lo:hi]).all()


def test_gather_commands_layout(tmp_path):
    import torch.multiprocessing as mp
    mp.spawn(_gather_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    for rank in range(2):
        g = np.load(os.path.join(str(tmp_path), "g%d.npy" % rank))
        assert g.shape == (2, 5, 3)
        assert (g[0] == 0).all() and (g[1] == 1).all()


def _gather_worker(rank, world, port, out_dir):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    out = sharding.gather_commands(torch.full((5, 3), float(rank), dtype=torch.float64))
    np.save(os.path.join(out_dir, "g%d.npy" % rank), out.numpy())
    dist.destroy_process_group()
