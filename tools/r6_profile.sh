#!/bin/bash
# one gpurun call: GPU tests, rocprofv3 evidence for C2 / C3 / C5 (kernel stats + PMC in separate passes) and K3, the default
# bench line (other_workloads, warm_tick, pcie variants ride in it), parity report, closed loop, tick latency
# usage: bash tools/r6_profile.sh <tag>      then: bash tools/r6_install.sh <tag> <prefix>
TAG=${1:-r06a}
O=gpurun_out/final_$TAG
mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -15 > $O/gpu_tests.log
# (first the calibration of FETCH_SIZE / WRITE_SIZE on K1's access shapes: profile_all.sh applies it to the traffic entries)
timeout 600 bash tools/profile_k1_traffic.sh $TAG > $O/profile_k1cal.txt 2>&1
timeout 900 bash tools/profile_all.sh $TAG C2 50 > $O/profile_c2.txt 2>&1
timeout 900 bash tools/profile_all.sh $TAG C3 4 > $O/profile_c3.txt 2>&1
timeout 900 bash tools/profile_all.sh $TAG C5 4 > $O/profile_c5.txt 2>&1
timeout 900 bash tools/profile_all.sh $TAG C4 4 > $O/profile_c4.txt 2>&1
python - <<PY
import json
out = {}
for d in ("gpurun_out/prof_$TAG", "gpurun_out/prof_${TAG}_C3", "gpurun_out/prof_${TAG}_C5", "gpurun_out/prof_${TAG}_C4"):
    try: out.update(json.load(open(d + "/traffic_entry.json")))
    except Exception as e: print("no traffic entry in", d, e)
json.dump(out, open("profiles/hbm_traffic.json", "w"), indent=1)
PY
timeout 900 python bench.py > $O/bench_c2.log 2>&1; tail -1 $O/bench_c2.log > $O/bench_c2.json
# (the A/B partner of the routed headline kernel: the dense direction for every instance, round 5's AUTO)
timeout 300 python bench.py --no-cpu-baseline --no-pcie --no-others --method 2 2>/dev/null | tail -1 > $O/bench_c2_dense_only.json
NEO_MPC_LIB=neo_mpc_planner2_amd/libneo_mpc_timing.so timeout 300 python tools/phase_timing_routed.py > $O/phase_timing_routed.txt 2>&1
timeout 600 bash tools/profile_k3.sh $TAG > $O/profile_k3.txt 2>&1
timeout 900 python tools/parity_report.py > $O/parity_report.txt 2>&1
timeout 300 python tools/bench_fleet_loop.py 2>/dev/null | tail -1 > $O/fleet_loop.json
timeout 300 python tools/bench_fleet_loop.py --pool 2>/dev/null | tail -1 > $O/fleet_loop_pool.json
timeout 120 python tools/bench_tick_latency.py 2>/dev/null | tail -1 > $O/tick_latency.json
timeout 600 python tools/bench_host_path.py 2>/dev/null | tail -1 > $O/host_path.json
timeout 300 python tools/bench_split_call.py 2>/dev/null | tail -1 > $O/split_call.json
timeout 300 python tools/bench_balance.py 2>/dev/null | tail -1 > $O/balanced_dispatch.json
timeout 600 python tools/soak_fleet.py 2>/dev/null | tail -1 > $O/soak_fleet.json
{ echo "== stage-wise branch (instances with a wall in reach)"; python tools/opcode_histogram.py k_solve_routedILi4ELb1ELi1024E --branch stagewise;
  echo; echo "== dense branch"; python tools/opcode_histogram.py k_solve_routedILi4ELb1ELi1024E --branch dense; } > $O/opcodes_c2.txt 2>&1
python tools/opcode_histogram.py k_solveILi4ELi3ELi1ELb1ELi1024E > $O/opcodes_c2_dense_only.txt 2>&1
python tools/opcode_histogram.py k_solveILi4ELi0ELi2ELb1ELi0E > $O/opcodes_riccati.txt 2>&1
cat $O/gpu_tests.log; cut -c1-1500 $O/bench_c2.json; grep -v amdgpu $O/parity_report.txt | tail -30; cat $O/fleet_loop.json | cut -c1-400; cat $O/tick_latency.json
