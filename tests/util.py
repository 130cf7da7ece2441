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



def closed_loop_on_the_mirror(params, cmap, probs, ticks, hz=30.0):
    """The fleet loop of neo_mpc_planner2_amd/fleet.py on the CPU mirror (oracle/mpc_oracle.c part 2): robots moved by
    their own commands, the carrot kept on its initial world bearing, warm start = the reference's shift.  Returns the
    per-tick command records."""
    from neo_mpc_planner2_amd import synthetic
    from oracle import c_oracle

    def yaw_of(q):
        return np.arctan2(2 * (q[:, 3] * q[:, 2] + q[:, 0] * q[:, 1]), 1 - 2 * (q[:, 1] ** 2 + q[:, 2] ** 2))
    probs = probs.copy()
    st, warm = synthetic.make_states(probs, params["control_steps"])
    yaw, pos = yaw_of(probs["cur_q"]).copy(), probs["cur_xy"].copy()
    c, s = np.cos(yaw), np.sin(yaw)
    off = np.stack([c * probs["carrot_xy"][:, 0] - s * probs["carrot_xy"][:, 1],
                    s * probs["carrot_xy"][:, 0] + c * probs["carrot_xy"][:, 1]], 1)
    carrot_yaw_w = yaw + yaw_of(probs["carrot_q"])
    probs["control_interval"] = 1.0 / hz
    probs["delta_t"] = 1.0 / hz
    out = []
    for _ in range(ticks):
        cm = c_oracle.solve_batch(params, cmap, probs, st, warm)[0]
        v = cm["vel"]
        yaw = yaw + v[:, 2] / hz
        c, s = np.cos(yaw), np.sin(yaw)
        pos = pos + np.stack([c * v[:, 0] - s * v[:, 1], s * v[:, 0] + c * v[:, 1]], 1) / hz
        probs["cur_xy"], probs["cur_q"] = pos, synthetic.yaw_quat(yaw)
        probs["carrot_xy"][:, 0] = c * off[:, 0] + s * off[:, 1]
        probs["carrot_xy"][:, 1] = -s * off[:, 0] + c * off[:, 1]
        probs["carrot_q"] = synthetic.yaw_quat(carrot_yaw_w - yaw)
        probs["cur_vel"] = v
        out.append(cm)
    return out


orc = orc  # re-export: tests use util.orc.make_params
