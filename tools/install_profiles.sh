#!/bin/bash
# copy the summaries of `bash tools/final_refresh.sh <tag>` (gpurun_out/) into profiles/ as r01_<letter>_*
# usage: bash tools/install_profiles.sh <tag, e.g. r01u> <prefix, e.g. r01_u> [old prefix to remove]
set -e
TAG=$1; T=$2; OLD=${3:-}
F=gpurun_out/final_$TAG; P=gpurun_out/prof_$TAG
[ -n "$OLD" ] && git rm -q --ignore-unmatch profiles/${OLD}_*
cp $P/kernels.txt profiles/${T}_kernels_c2.txt
cp $P/pmc.txt profiles/${T}_pmc_c2.txt
cp $F/bench_c2.json profiles/${T}_bench_c2.json
cp $F/parity_report.txt profiles/${T}_parity_report.txt
cp $F/other_configs.jsonl profiles/${T}_other_configs.jsonl
cp $F/batch_scaling.txt profiles/${T}_batch_scaling.txt
cp $F/carrot_hbm.jsonl profiles/${T}_carrot_hbm.jsonl
[ -f $F/fleet_loop.json ] && cp $F/fleet_loop.json profiles/${T}_fleet_loop.json
[ -f $F/fleet_loop_pool.json ] && cp $F/fleet_loop_pool.json profiles/${T}_fleet_loop_pool.json
python - <<PY
import json, re
pmc = open("profiles/${T}_pmc_c2.txt").read()
f = float(re.search(r"FETCH_SIZE\s+([\d.]+)", pmc).group(1)); w = float(re.search(r"WRITE_SIZE\s+([\d.]+)", pmc).group(1))
k = re.search(r"(k_solve<[^>]*>)", open("profiles/${T}_kernels_c2.txt").read()).group(1)
v = float(re.search(r"SQ_INSTS_VALU\s+([\d.]+)", pmc).group(1))
json.dump({"C2": (f + w) * 1024, "C2_valu_insts": v, "note": "(FETCH_SIZE %.1f KB + WRITE_SIZE %.1f KB) * 1024 per %s launch (4096 instances), rocprofv3 --pmc in separate passes (tools/profile_all.sh $TAG), raw counters: K1's loads are 8-byte and dword accesses, so the gfx950 x2 FETCH_SIZE correction for 16 B/lane streams is not applied; profiles/${T}_pmc_c2.txt. Algorithmic bytes 885 B x 4096 = 3.62 MB." % (f, w, k)}, open("profiles/hbm_traffic.json", "w"))
open("profiles/hbm_traffic.json", "a").write("\n")
PY
ls profiles
