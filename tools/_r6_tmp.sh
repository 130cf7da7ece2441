O=gpurun_out/check_r06j; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -60 > $O/gpu_tests.log
timeout 900 python bench.py --no-cpu-baseline --no-pcie > $O/bench_c2.log 2>&1; tail -1 $O/bench_c2.log > $O/bench_c2.json
cat $O/gpu_tests.log
python - $O/bench_c2.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
print({k:d.get(k) for k in ("value","ms_per_step")}, d["roofline"].get("kernel_ms"), d.get("solver"), d.get("two_streams"))
for o in d.get("other_workloads",[]): print(o.get("workload"), o.get("value"), o.get("kernel_ms"), o.get("solver"))
print("warm_tick", {k:v for k,v in d.get("warm_tick",{}).items() if k not in ("what","launch_order")})
PY
