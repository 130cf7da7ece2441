#!/usr/bin/env python3
"""Turn rocprofv3 outputs (rocpd .db from --kernel-trace --stats, *_counter_collection.csv from
--pmc) into the small text summaries committed under profiles/.

    python tools/rocprof_summary.py kernels gpurun_out/prof/x_results.db > profiles/rNN_kernels.txt
    python tools/rocprof_summary.py pmc gpurun_out/pmc_dir > profiles/rNN_pmc.txt
"""
import collections
import csv
import glob
import sqlite3
import sys


def kernels(db):
    c = sqlite3.connect(db)
    print("# rocprofv3 --kernel-trace --stats summary of %s" % db)
    print("%-90s %6s %12s %10s %10s %10s %8s %8s %6s" % ("kernel", "calls", "total_us", "avg_us", "min_us", "max_us",
                                                          "lds_B", "scratch", "vgpr"))
    q = ("select name, count(*), sum(duration), avg(duration), min(duration), max(duration), avg(lds_size), "
         "avg(scratch_size), avg(vgpr_count + accum_vgpr_count) from kernels group by name order by sum(duration) desc")
    for r in c.execute(q):
        print("%-90s %6d %12.1f %10.2f %10.2f %10.2f %8d %8d %6d" % (r[0][:90], r[1], r[2] / 1e3, r[3] / 1e3,
                                                                   r[4] / 1e3, r[5] / 1e3, r[6], r[7], r[8]))


def pmc(directory, match="k_solve"):
    print("# rocprofv3 --pmc summary (per-dispatch mean over dispatches of kernels matching %r) of %s"
          % (match, directory))
    for f in sorted(glob.glob(directory + "/*_counter_collection.csv")):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if match in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in sorted(agg.items()):
            print("%-28s %18.1f   (n=%d, pass %s)" % (k, sum(v) / len(v), len(v), f.split("/")[-1]))


if __name__ == "__main__":
    {"kernels": kernels, "pmc": pmc}[sys.argv[1]](*sys.argv[2:])
