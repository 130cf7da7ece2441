#!/bin/bash
# study: GPU-side timeline of bench.py with the RCCL path forced on one rank: gaps between K1 launches
R=$PWD; OUT=$R/gpurun_out/trace_gaps; rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
NEO_MPC_BENCH_FORCE_DIST=${FORCE:-1} rocprofv3 --kernel-trace --output-format csv -d $OUT -o t -- python $R/bench.py --steps 40 --warmup 5 --no-cpu-baseline > $OUT/log.txt 2>&1
cd $R
python - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/trace_gaps/**/t_kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ks = [r for r in rows if "k_solve" in r["Kernel_Name"]]
names = collections.Counter(r["Kernel_Name"][:60] for r in rows)
print(names.most_common(8))
st = [int(r["Start_Timestamp"]) for r in ks]; en = [int(r["End_Timestamp"]) for r in ks]
gaps = [(st[i + 1] - en[i]) / 1e3 for i in range(len(ks) - 1)]
dur = [(e - s) / 1e3 for s, e in zip(st, en)]
import statistics
print("k_solve launches %d: duration median %.1f us; gap to next launch median %.1f us (min %.1f max %.1f); period median %.1f us" % (
    len(ks), statistics.median(dur), statistics.median(gaps[-30:]), min(gaps[-30:]), max(gaps[-30:]), statistics.median([(st[i+1]-st[i])/1e3 for i in range(len(st)-31, len(st)-1)])))
# what runs inside the gaps
others = [r for r in rows if "k_solve" not in r["Kernel_Name"] and int(r["Start_Timestamp"]) > st[-20]]
for r in others[:12]:
    print("  %-60s start+%.1f us dur %.1f us" % (r["Kernel_Name"][:60], (int(r["Start_Timestamp"]) - st[-20]) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
PY
