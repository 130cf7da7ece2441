"""CPU checks of the C-ABI boundary: libneo_mpc.so loads without a GPU and exports every
symbol include/neo_mpc.h declares; the Python record layouts equal the C structs (checked with
a gcc-compiled probe); a handle cannot be created without a device (no CPU fallback)."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

from neo_mpc_planner2_amd import _lib, abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "neo_mpc.h")


def declared_functions():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(neo_mpc_[a-z_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    names = declared_functions()
    assert len(names) >= 14
    for n in names:
        assert hasattr(lib, n), n
    assert sorted(_lib.EXPORTS) == names
    assert lib.neo_mpc_abi_version() == abi.ABI_VERSION == 2
    lib.neo_mpc_behaviour_version.restype = C.c_int
    assert lib.neo_mpc_behaviour_version() == 6
    header = open(HEADER).read()
    assert "#define NEO_MPC_ABI_VERSION 2" in header and "#define NEO_MPC_BEHAVIOUR_VERSION 6" in header


def test_record_layouts_match_the_header(tmp_path):
    src = tmp_path / "probe.c"
    src.write_text(r'''
#include <stdio.h>
#include <stddef.h>
#include "neo_mpc.h"
#define P(t, f) printf(#t "." #f " %zu\n", offsetof(t, f))
int main(void) {
  printf("sizeof.params %zu\nsizeof.problem %zu\nsizeof.state %zu\nsizeof.command %zu\nsizeof.batch %zu\n",
         sizeof(neo_mpc_params), sizeof(neo_mpc_problem), sizeof(neo_mpc_state), sizeof(neo_mpc_command),
         sizeof(neo_mpc_batch));
  P(neo_mpc_params, control_steps); P(neo_mpc_params, step_tolerance); P(neo_mpc_params, kink_radius); P(neo_mpc_params, stall_step); P(neo_mpc_params, method); P(neo_mpc_params, window_tolerance);
  P(neo_mpc_problem, carrot_xy); P(neo_mpc_problem, goal_xyz); P(neo_mpc_problem, cur_vel);
  P(neo_mpc_problem, control_interval); P(neo_mpc_problem, footprint_cost); P(neo_mpc_problem, map_index);
  P(neo_mpc_problem, switch_opt); P(neo_mpc_problem, skip);
  P(neo_mpc_state, old_goal); P(neo_mpc_state, waiting_time); P(neo_mpc_state, has_old_goal);
  P(neo_mpc_state, collision_footprint); P(neo_mpc_state, has_prev_u0); P(neo_mpc_state, prev_u0);
  P(neo_mpc_command, cost); P(neo_mpc_command, status); P(neo_mpc_command, flags);
  P(neo_mpc_batch, footprints); P(neo_mpc_batch, footprint_points); P(neo_mpc_batch, velocities);
  return 0;
}''')
    exe = tmp_path / "probe"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    got = dict(line.split() for line in subprocess.check_output([str(exe)]).decode().splitlines())
    got = {k: int(v) for k, v in got.items()}
    assert got["sizeof.params"] == C.sizeof(abi.NeoMpcParams)
    assert got["sizeof.problem"] == abi.PROBLEM_DTYPE.itemsize == 256
    assert got["sizeof.state"] == abi.STATE_DTYPE.itemsize == 128
    assert got["sizeof.command"] == abi.COMMAND_DTYPE.itemsize == 48
    assert got["sizeof.batch"] == C.sizeof(abi.NeoMpcBatch)
    for f in ("control_steps", "step_tolerance", "kink_radius", "stall_step", "method", "window_tolerance"):
        assert got["neo_mpc_params." + f] == getattr(abi.NeoMpcParams, f).offset
    for f in ("carrot_xy", "goal_xyz", "cur_vel", "control_interval", "footprint_cost", "map_index", "switch_opt", "skip"):
        assert got["neo_mpc_problem." + f] == abi.PROBLEM_DTYPE.fields[f][1]
    for f in ("old_goal", "waiting_time", "has_old_goal", "collision_footprint", "has_prev_u0", "prev_u0"):
        assert got["neo_mpc_state." + f] == abi.STATE_DTYPE.fields[f][1]
    # (round 5: the build's one hint lives where ABI 1 had reserved bytes -- the last 28 of the record, same size)
    assert got["neo_mpc_state.has_prev_u0"] == 100 and got["neo_mpc_state.prev_u0"] == 104
    for f in ("cost", "status", "flags"):
        assert got["neo_mpc_command." + f] == abi.COMMAND_DTYPE.fields[f][1]
    for f in ("footprints", "footprint_points", "velocities"):
        assert got["neo_mpc_batch." + f] == getattr(abi.NeoMpcBatch, f).offset


def test_default_params_are_the_reference_nodes_declared_defaults():
    """neo_mpc_default_params == mpc_optimization_server.py:49-75 (no GPU needed)."""
    from neo_mpc_planner2_amd.mpc_optimization_server import DEFAULT_PARAMS
    lib = _lib.load()
    p = abi.NeoMpcParams()
    assert lib.neo_mpc_default_params(C.byref(p)) == 0
    for k, v in DEFAULT_PARAMS.items():
        assert getattr(p, k) == v, k
    assert p.compat_flags == abi.COMPAT_ODOM_YAW_GOAL_W and p.max_iterations == 100


@pytest.mark.skipif(os.path.exists("/dev/kfd"), reason="a GPU is present")
def test_no_cpu_fallback():
    """Without a device the product refuses to run -- it never routes through oracle/ or any
    CPU path."""
    from neo_mpc_planner2_amd.solver import BatchSolver
    with pytest.raises(_lib.NeoMpcError) as e:
        BatchSolver({})
    assert "no CPU fallback" in str(e.value)


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "neo_mpc_planner2_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in text and "from oracle" not in text, f
                assert "oracle/" not in text.replace("oracle/mpc_oracle.c, OpenMP", ""), f


def test_request_record_maps_the_service_fields():
    from neo_mpc_planner2_amd import mpc_optimization_server as srv
    req = srv.make_request(cur_xy=(1, 2), cur_q=(0, 0, 0.6, 0.8), carrot_xy=(0.3, -0.2), carrot_q=(0, 0, 0.1, 0.99),
                           goal_xyz=(5, 6, 0.5), goal_q=(0, 0, 1, 0), cur_vel=(0.1, 0.2, 0.3), control_interval=0.05)
    r = srv.request_record(req, delta_t=0.25)[0]
    assert tuple(r["cur_xy"]) == (1, 2) and tuple(r["cur_q"]) == (0, 0, 0.6, 0.8)
    assert tuple(r["carrot_xy"]) == (0.3, -0.2) and tuple(r["goal_xyz"]) == (5, 6, 0.5)
    assert tuple(r["goal_q"]) == (0, 0, 1, 0) and tuple(r["cur_vel"]) == (0.1, 0.2, 0.3)
    assert r["control_interval"] == 0.05 and r["delta_t"] == 0.25


def test_yaml_parameter_block_uses_the_reference_names(tmp_path):
    from neo_mpc_planner2_amd import mpc_optimization_server as srv
    y = tmp_path / "nav.yaml"
    y.write_text("mpc_optimization_server:\n  ros__parameters:\n    acc_x_limit: 2.5\n    w_trans: 0.82\n"
                 "    opt_tolerance: 1e-3\n    prediction_horizon: 0.8\n    control_steps: 3\n    unknown_key: 1\n")
    p = srv.load_params_yaml(str(y))
    assert p["acc_x_limit"] == 2.5 and p["w_trans"] == 0.82 and p["prediction_horizon"] == 0.8
    assert float(p["opt_tolerance"]) == 1e-3 and p["max_vel_x"] == 0.5 and "unknown_key" not in p
    assert set(srv.DEFAULT_PARAMS) == set(abi.ROS_PARAM_NAMES)


def test_plugin_seam_example_compiles_and_links(tmp_path):
    """examples/plugin_seam.cpp (the cpp:240-252 replacement of INTEGRATION.md) builds against
    include/neo_mpc.h and links libneo_mpc.so; running it needs no GPU (it only checks the ABI)."""
    exe = tmp_path / "seam"
    subprocess.check_call(["g++", "-std=c++17", "-Wall", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", "plugin_seam.cpp"),
                           "-L", os.path.join(ROOT, "neo_mpc_planner2_amd"), "-lneo_mpc",
                           "-Wl,-rpath," + os.path.join(ROOT, "neo_mpc_planner2_amd"),
                           "-Wl,-rpath,/opt/rocm/lib", "-o", str(exe)])
    assert subprocess.call([str(exe)]) == 0


def test_bench_refuses_to_run_more_ranks_than_devices():
    """`python bench.py --gpus N` without a launcher spawns its own N ranks -- and says so loudly when the node has
    fewer devices (here: none) instead of quietly running on one."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("needs a box with fewer than 2 GPUs")
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                         capture_output=True, text=True, timeout=300, cwd=root)
    assert out.returncode != 0 and "HIP device(s) visible" in (out.stderr + out.stdout)


def test_cb_params_accepts_rclpy_style_enum_types():
    """rclpy's Parameter.Type is a plain Enum (DOUBLE.value == 3), not an int: such parameters must be applied
    (py:407 compares against the enum member), integers and other types skipped."""
    import enum
    from types import SimpleNamespace as NS
    from neo_mpc_planner2_amd import mpc_optimization_server as srv

    class Type(enum.Enum):
        INTEGER = 2
        DOUBLE = 3

    node = object.__new__(srv.MpcOptimizationServer)      # no GPU needed: only cb_params' bookkeeping
    node.reference_quirks = True
    node._params = dict(srv.README_PARAMS)
    applied = {}
    node._solver = NS(set_params=lambda **kw: applied.update(kw))
    node.cb_params([NS(name="w_trans", value=0.4, type_=Type.DOUBLE), NS(name="w_orient", value=7, type_=Type.INTEGER),
                    NS(name="w_control", value=0.2, type_=3)])
    assert node.w_trans == 0.4 and node.w_control == 0.2 and not hasattr(node, "w_orient")
    assert applied == {"w_trans": 0.4, "w_control": 0.2}


def test_bench_cpu_worker_and_cpu_count_without_a_gpu():
    """The all-cores CPU leg of bench.py: a worker is a fresh interpreter that never touches torch or the HIP
    library (it runs beside a live HIP runtime on the GPU box), and the process count respects the cgroup quota."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--cpu-worker", "1:4:0.5"],
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-500:]
    rec = json.loads(out.stdout.strip().splitlines()[-1])
    assert rec["done"] > 0 and rec["rate"] > 0.0
    sys.path.insert(0, root)
    import bench
    n = bench.usable_cpus()
    assert 1 <= n <= (os.cpu_count() or 1)


def test_baseline_kernels_are_scratch_free():
    """The kernels of the BASELINE configs -- k_solve_routed<4, tame, static tile> (config 2 / 4: control_steps 3, AUTO) and
    k_solve<4, 0, stage-wise, tame> (configs 3 and 5) -- spill no vector register: 0 bytes of scratch per lane at their four
    waves per SIMD (round 5's config-3 / 5 kernel had picked up 24 bytes per lane unnoticed: its HBM writes doubled).  The
    compiler's own resource remarks for the translation unit both live in, with the flags of the Makefile; no GPU needed."""
    import os
    import re
    import subprocess
    csrc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "neo_mpc_planner2_amd", "csrc")
    mk = open(os.path.join(csrc, "Makefile")).read()
    flags = re.search(r"^RICCATI_FLAGS := (.*)$", mk, re.M).group(1).split()
    out = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-function",
                          "-Wno-pass-failed"] + flags + ["-Rpass-analysis=kernel-resource-usage", "-x", "hip", "-c",
                          "neo_mpc_riccati.hip", "-o", os.devnull], cwd=csrc, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    rows, cur = {}, None
    for line in out.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = m.group(1)
            rows[cur] = {}
            continue
        m = re.search(r"remark:\s+([A-Za-z ]+)(?: \[[^\]]*\])?: (\d+)", line)
        if m and cur:
            rows[cur][m.group(1).strip()] = int(m.group(2))
    wanted = {"k_solve_routedILi4ELb1ELi1024E": "config 2 / 4", "7k_solveILi4ELi0ELi2ELb1ELi0E": "configs 3, 5"}
    for key, what in wanted.items():
        hit = [v for k, v in rows.items() if key in k]
        assert len(hit) == 1, (key, list(rows))
        assert hit[0]["ScratchSize"] == 0 and hit[0]["VGPRs Spill"] == 0 and hit[0]["VGPRs"] <= 128 and hit[0]["Occupancy"] >= 4, (what, hit[0])
