#!/bin/bash
# one gpurun call: GPU tests, the default bench line, rocprofv3 evidence for C2 / C3 / C5 and K3, parity report
# usage: bash tools/r2_profile.sh <tag>
TAG=${1:-r02a}
O=gpurun_out/final_$TAG
mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -15 > $O/gpu_tests.log
timeout 900 bash tools/profile_all.sh $TAG C2 50 > $O/profile_c2.txt 2>&1
timeout 900 bash tools/profile_all.sh $TAG C3 4 > $O/profile_c3.txt 2>&1
timeout 900 bash tools/profile_all.sh $TAG C5 4 > $O/profile_c5.txt 2>&1
# the PMC passes above are of THIS build: the bench lines below report them (tools/install_profiles.sh writes the
# same file into the repo afterwards)
python - <<PY
import json
out = {}
for d in ("gpurun_out/prof_$TAG", "gpurun_out/prof_${TAG}_C3", "gpurun_out/prof_${TAG}_C5"):
    try: out.update(json.load(open(d + "/traffic_entry.json")))
    except Exception as e: print("no traffic entry in", d, e)
json.dump(out, open("profiles/hbm_traffic.json", "w"), indent=1)
PY
timeout 600 python bench.py > $O/bench_c2.log 2>&1; tail -1 $O/bench_c2.log > $O/bench_c2.json
timeout 600 bash tools/profile_k3.sh $TAG > $O/profile_k3.txt 2>&1
timeout 900 python tools/parity_report.py > $O/parity_report.txt 2>&1
{
  timeout 300 python bench.py --workload C3 --steps 5 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1
  timeout 300 python bench.py --workload C4 --steps 5 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1
  timeout 300 python bench.py --workload C5 --steps 5 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1
  timeout 120 python tools/bench_tick_latency.py 2>/dev/null | tail -1
} > $O/other_configs.jsonl
cat $O/gpu_tests.log; cat $O/bench_c2.json; grep -v amdgpu $O/parity_report.txt
python - <<PY
import json
for l in open("$O/other_configs.jsonl"):
    try: d=json.loads(l)
    except Exception: continue
    if isinstance(d, dict) and isinstance(d.get("config"), dict) and "workload" in d["config"]:
        print(d["config"]["workload"][:40], "| %.4g solves/s | kernel %.3f ms | iters %.2f" % (d["value"], d["roofline"]["kernel_ms"], d["solver"]["mean_iterations"]))
    else: print(l.strip()[:300])
PY
