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

orc = orc  # re-export: tests use util.orc.make_params
