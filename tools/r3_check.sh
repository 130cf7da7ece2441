#!/bin/bash
# one gpurun call of round 3: GPU tests, the default bench line (with other_workloads / warm_tick / pcie variants),
# the closed-loop tool.   usage: bash tools/r3_check.sh <tag> [pytest -k expression]
TAG=${1:-r03a}
O=gpurun_out/$TAG
mkdir -p $O
if [ -n "$2" ]; then
  timeout 2400 python -m pytest tests -m gpu -q -k "$2" 2>&1 | tail -40 > $O/gpu_tests.log
else
  timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -40 > $O/gpu_tests.log
fi
timeout 900 python bench.py > $O/bench_c2.log 2>&1; tail -1 $O/bench_c2.log > $O/bench_c2.json
timeout 300 python tools/bench_fleet_loop.py 2>/dev/null | tail -1 > $O/fleet_loop.json
cat $O/gpu_tests.log
python - <<PY
import json
d = json.load(open("$O/bench_c2.json"))
print("C2 %.4g solves/s  ms/step %.4f  kernel %.4f ms  it %.2f max %d" % (d["value"], d["ms_per_step"], d["roofline"]["kernel_ms"], d["solver"]["mean_iterations"], d["solver"]["max_iterations_seen"]))
pc = d.get("pcie_inclusive", {})
print("pcie pageable %.4g (%.3f ms)" % (pc.get("value", 0), pc.get("ms_per_call", 0)))
for k in ("pinned", "pinned_zerocopy_out", "pinned_staged", "pinned_two_in_flight"):
    r = pc.get(k, {})
    print(" ", k, r.get("value"), r.get("ms_per_call"), r.get("commands_identical"), r.get("error"))
for o in d.get("other_workloads", []):
    if "error" in o: print(o); continue
    print("%-18s %.4g solves/s kernel %.3f ms it %.2f max %d capped %d" % (o["workload"], o["value"], o["kernel_ms"], o["solver"]["mean_iterations"], o["solver"]["max_iterations_seen"], o["solver"]["status_max_iter"]))
print("warm_tick", d.get("warm_tick"))
print("cpu", {k: d.get(k, {}).get("value") for k in ("cpu_baseline", "cpu_mirror")})
print("fleet", open("$O/fleet_loop.json").read()[:600])
PY
