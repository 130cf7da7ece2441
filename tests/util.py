"""Shared helpers for the parity tests: fixture loading and record <-> oracle glue."""
import os

import numpy as np

from neo_mpc_planner2_amd.abi import PROBLEM_DTYPE
from oracle import mpc_oracle as orc

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def params_from(keys, vec):
    p = {str(k): float(v) for k, v in zip(keys, vec)}
    p["control_steps"] = int(p["control_steps"])
    return p


def problems_from(raw):
    return np.ascontiguousarray(raw).view(PROBLEM_DTYPE).reshape(raw.shape[:-1])


def oracle_problem(row, footprint=None):
    fp = ()
    if footprint is not None and not np.isnan(footprint).any():
        fp = [tuple(p) for p in footprint]
    return orc.Problem(row["cur_xy"], row["cur_q"], row["carrot_xy"], row["carrot_q"],
                       row["goal_xyz"], row["goal_q"], row["cur_vel"],
                       control_interval=float(row["control_interval"]),
                       delta_t=float(row["delta_t"]), footprint=fp)


def oracle_costmap(cells, meta):
    return orc.Costmap(cells, meta[0], meta[1], meta[2])



def load_seam_dump(path):
    """`examples/plugin_seam.cpp --run --dump FILE`: header (tick count, control_steps, map size, parameters, map
    geometry, the costmap versions) and one record per control tick -- what the C++ seam sent, held and got."""
    import ctypes as C
    from neo_mpc_planner2_amd import abi
    raw = open(path, "rb").read()
    assert raw[:8] == b"NEOSEAM1", raw[:8]
    ticks, n_steps, size, n_maps = np.frombuffer(raw, dtype="<i4", count=4, offset=8)
    off = 24
    ps = abi.NeoMpcParams.from_buffer_copy(raw[off:off + C.sizeof(abi.NeoMpcParams)])
    off += C.sizeof(abi.NeoMpcParams)
    params = {name: getattr(ps, name) for name, _ in abi.NeoMpcParams._fields_ if name != "reserved_i"}
    geom = np.frombuffer(raw, dtype="<f8", count=3, offset=off)
    off += 24
    maps = np.frombuffer(raw, dtype=np.uint8, count=n_maps * size * size, offset=off).reshape(n_maps, size, size)
    off += n_maps * size * size
    nv = 3 * int(n_steps)
    rec = np.dtype([("map", "<i4"), ("tick", "<i4"), ("problem", abi.PROBLEM_DTYPE), ("state_before", abi.STATE_DTYPE),
                    ("warm_before", "<f8", (nv,)), ("command", abi.COMMAND_DTYPE), ("state_after", abi.STATE_DTYPE),
                    ("warm_after", "<f8", (nv,))])
    assert len(raw) - off == int(ticks) * rec.itemsize, (len(raw) - off, int(ticks), rec.itemsize)
    return params, tuple(geom), maps, np.frombuffer(raw, dtype=rec, count=int(ticks), offset=off)



def solve_group(name, prefix):
    """One `_g3_group` of a cold-solve fixture (oracle/gen_golden.py): (arrays by bare name, params, problems,
    has_map mask)."""
    g = load(name)
    params = params_from(g["param_keys"], g[prefix + "params"])
    group = {k[len(prefix):]: g[k] for k in g.files if k.startswith(prefix)} if prefix else {k: g[k] for k in g.files}
    probs = problems_from(group["problems"])
    hm = group["has_map"].astype(bool) if "has_map" in group else np.zeros(len(probs), dtype=bool)
    return group, params, probs, hm


#: (fixture, prefix) of every cold-solve group made at parameters other than the README's at opt_tolerance 1e-3
G9_GROUPS = (("g9_solves_pydefaults.npz", "n3_"), ("g9_solves_pydefaults.npz", "n8_"))
G8_MID_GROUPS = tuple(("g8_mid.npz", "r%02d_" % r) for r in (10, 15, 20, 25, 30))
EPISODE_FIXTURES = ("g4_episodes.npz", "g4_episodes_n8.npz", "g4_episodes_params.npz", "g9_episodes_pydefaults.npz")


orc = orc  # re-export: tests use util.orc.make_params
