#!/bin/bash
# K3 (k_ingest, the HBM-streaming costmap kernel): rocprofv3 kernel stats + FETCH_SIZE / WRITE_SIZE in separate
# passes for a pool of 64 maps of 1000x1000 (64 MB read, 84 MB written: beyond L2, inside the 256 MB Infinity
# Cache) and of 4096 windows of 200x200.
# usage: bash tools/profile_k3.sh <tag>   -> gpurun_out/prof_<tag>_k3/
set -u
TAG=${1:-r02}
R=$PWD
OUT=$R/gpurun_out/prof_${TAG}_k3
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for cfg in "64 1000" "4096 200"; do
  set -- $cfg
  n=m$1_s$2
  python $R/tools/bench_ingest.py $1 $2 2>/dev/null | tail -1 > $OUT/bench_$n.json
  rocprofv3 --kernel-trace --stats -d $OUT -o trace_$n -- python $R/tools/bench_ingest.py $1 $2 > $OUT/trace_$n.log 2>&1
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_$n -o pmc_$c -- python $R/tools/bench_ingest.py $1 $2 > $OUT/pmc_${n}_$c.log 2>&1
  done
  {
    echo "## k_ingest, $1 maps of $2 x $2"
    cat $OUT/bench_$n.json
    python $R/tools/rocprof_summary.py kernels $OUT/trace_${n}_results.db | grep -E "^#|^kernel|k_ingest"
    python $R/tools/rocprof_summary.py pmc $OUT/pmc_$n k_ingest
    python - <<PY
import json, re, subprocess
b = json.load(open("$OUT/bench_$n.json"))
txt = subprocess.run(["python", "$R/tools/rocprof_summary.py", "pmc", "$OUT/pmc_$n", "k_ingest"], capture_output=True, text=True).stdout
f = float(re.search(r"FETCH_SIZE\s+([\d.]+)", txt).group(1)) * 1024
w = float(re.search(r"WRITE_SIZE\s+([\d.]+)", txt).group(1)) * 1024
print("calibration: FETCH_SIZE %.1f MB raw vs %.1f MB of cells read once (ratio %.2f); WRITE_SIZE %.1f MB raw vs %.1f MB written (ratio %.2f)"
      % (f / 1e6, b["read_bytes"] / 1e6, f / b["read_bytes"], w / 1e6, b["written_bytes"] / 1e6, w / b["written_bytes"]))
PY
  } > $OUT/summary_$n.txt 2>&1
done
cd $R
cat $OUT/summary_*.txt > $OUT/k3.txt
cat $OUT/k3.txt
