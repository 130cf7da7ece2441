#!/bin/bash
# Calibrates rocprofv3's FETCH_SIZE / WRITE_SIZE on K1's own access shapes (tools/mb_k1_traffic.hip), in separate --pmc
# passes, like tools/profile_k3.sh does for K3.   usage: bash tools/profile_k1_traffic.sh <tag>
#   -> gpurun_out/prof_<tag>_k1cal/calibration.json  (copied to profiles/<tag>_k1_traffic_calibration.json by hand)
set -u
TAG=${1:-r04}
R=$PWD
OUT=$R/gpurun_out/prof_${TAG}_k1cal
mkdir -p $OUT
export TMPDIR=/tmp
[ -x $R/tools/_build/mb_k1_traffic ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 $R/tools/mb_k1_traffic.hip -o $R/tools/_build/mb_k1_traffic
cd /tmp
$R/tools/_build/mb_k1_traffic > $OUT/known_bytes.json
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $OUT -o pmc_$c -- $R/tools/_build/mb_k1_traffic > $OUT/pmc_$c.log 2>&1
done
cd $R
python - <<PY > $OUT/calibration.json
import collections, csv, glob, json
known = json.load(open("$OUT/known_bytes.json"))
vals = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        for k in ("probe_records", "probe_tile", "probe_write"):
            if k in r["Kernel_Name"]:
                vals[k][r["Counter_Name"]].append(float(r["Counter_Value"]) * 1024.0)   # (KB units)
mean = lambda v: sum(v) / len(v) if v else None
out = {"instances": known["instances"], "known": known, "raw_counter_bytes": {k: {c: mean(v) for c, v in d.items()} for k, d in vals.items()}}
rec, wr, tile = known["probe_records"], known["probe_write"], known["probe_tile"]
f_rec, f_tile, w_wr = mean(vals["probe_records"]["FETCH_SIZE"]), mean(vals["probe_tile"]["FETCH_SIZE"]), mean(vals["probe_write"]["WRITE_SIZE"])
out["fetch_ratio_records"] = f_rec / rec["arrays"]          # counter / bytes of the lines the loads touch (= the arrays)
out["fetch_ratio_records_vs_useful"] = f_rec / rec["useful"]
out["fetch_ratio_tile_vs_sectors"] = f_tile / tile["sectors64"]
out["write_ratio"] = w_wr / wr["arrays"]
out["write_ratio_vs_useful"] = w_wr / wr["useful"]
out["note"] = ("rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over tools/mb_k1_traffic.hip: K1's record loads "
               "(8 B per lane, 216 / 104 / 72-byte rows), its reach-tile loads (dwords) and K2's stores, a million instances each")
print(json.dumps(out, indent=1))
PY
cat $OUT/calibration.json
