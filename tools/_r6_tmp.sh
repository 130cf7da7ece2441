O=gpurun_out/check_r06e; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -150 > $O/gpu_tests.log
NEO_MPC_LIB=neo_mpc_planner2_amd/libneo_mpc_timing.so timeout 300 python tools/phase_timing_routed.py > $O/phase.txt 2>&1
timeout 900 python bench.py > $O/bench_c2.log 2>&1; tail -1 $O/bench_c2.log > $O/bench_c2.json
timeout 300 python bench.py --no-cpu-baseline --no-pcie --no-others --method 2 2>/dev/null | tail -1 > $O/ab_dense.json
timeout 300 python bench.py --no-cpu-baseline --no-pcie --no-others 2>/dev/null | tail -1 > $O/ab_routed.json
timeout 300 python tools/bench_fleet_loop.py 2>/dev/null | tail -1 > $O/fleet_loop.json
cat $O/gpu_tests.log; cat $O/phase.txt
python - $O/bench_c2.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
print({k:d.get(k) for k in ("value","ms_per_step")}, d["roofline"].get("kernel_ms"), d.get("solver"))
for o in d.get("other_workloads",[]): print(o.get("workload"), o.get("value"), o.get("kernel_ms"), o.get("solver"))
print("warm_tick", {k:v for k,v in d.get("warm_tick",{}).items() if k!="what"})
PY
for f in $O/ab_*.json; do echo $f; python - $f <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); print({k:d.get(k) for k in ("value","ms_per_step")}, d.get("roofline",{}).get("kernel_ms"), d.get("solver"))
except Exception as e: print("unreadable", e)
PY
done
cut -c1-700 $O/fleet_loop.json
