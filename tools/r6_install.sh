#!/bin/bash
# copy the summaries of `bash tools/r6_profile.sh <tag>` (gpurun_out/) into profiles/ as <prefix>_*
# usage: bash tools/r6_install.sh <tag, e.g. r04a> <prefix, e.g. r04_a> [old prefix to remove]
set -e
TAG=$1; T=$2; OLD=${3:-}
F=gpurun_out/final_$TAG; P=gpurun_out/prof_$TAG
[ -n "$OLD" ] && git rm -q --ignore-unmatch profiles/${OLD}_* || true
cp $P/kernels.txt profiles/${T}_kernels_c2.txt
cp $P/pmc.txt profiles/${T}_pmc_c2.txt
for w in C3 C5 C4; do
  lw=$(echo $w | tr A-Z a-z)
  cp ${P}_$w/kernels.txt profiles/${T}_kernels_$lw.txt
  cp ${P}_$w/pmc.txt profiles/${T}_pmc_$lw.txt
done
cp ${P}_k3/k3.txt profiles/${T}_k3_ingest.txt
cp $F/bench_c2.json profiles/${T}_bench_c2.json
grep -v amdgpu.ids $F/parity_report.txt > profiles/${T}_parity_report.txt
cp $F/fleet_loop.json profiles/${T}_fleet_loop.json
cp $F/fleet_loop_pool.json profiles/${T}_fleet_loop_pool.json
cp $F/tick_latency.json profiles/${T}_tick_latency.json
cp $F/gpu_tests.log profiles/${T}_gpu_tests.txt
cp $F/host_path.json profiles/${T}_host_path.json
cp $F/split_call.json profiles/${T}_split_call.json
cp $F/balanced_dispatch.json profiles/${T}_balanced_dispatch.json
cp $F/soak_fleet.json profiles/${T}_soak_fleet.json
cp $F/opcodes_c2.txt profiles/${T}_opcodes_c2.txt
cp $F/opcodes_c2_dense_only.txt profiles/${T}_opcodes_c2_dense_only.txt
cp $F/bench_c2_dense_only.json profiles/${T}_bench_c2_dense_only.json
grep -v amdgpu.ids $F/phase_timing_routed.txt > profiles/${T}_phase_timing_routed.txt
cp $F/opcodes_riccati.txt profiles/${T}_opcodes_riccati.txt
cp gpurun_out/prof_${TAG}_k1cal/calibration.json profiles/${T}_k1_traffic_calibration.json
python - <<PY
import json
out = {}
for d in ("$P", "${P}_C3", "${P}_C5", "${P}_C4"):
    out.update(json.load(open(d + "/traffic_entry.json")))
json.dump(out, open("profiles/hbm_traffic.json", "w"), indent=1)
open("profiles/hbm_traffic.json", "a").write("\n")
print({k: (v["hbm_bytes"], v["valu_insts"], v["source_sha"]) for k, v in out.items()})
PY
ls profiles | grep $T
