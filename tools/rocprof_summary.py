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
    print("# (vgpr_rp = vgpr_count + accum_vgpr_count as rocprofv3 records them -- on gfx950 half of what the compiler\n"
          "#  reports, `make -C neo_mpc_planner2_amd/csrc resource-usage` is the authoritative figure)")
    print("%-90s %6s %12s %10s %10s %10s %8s %8s %7s" % ("kernel", "calls", "total_us", "avg_us", "min_us", "max_us",
                                                          "lds_B", "scratch", "vgpr_rp"))
    q = ("select name, count(*), sum(duration), avg(duration), min(duration), max(duration), avg(lds_size), "
         "avg(scratch_size), avg(vgpr_count + accum_vgpr_count) from kernels group by name order by sum(duration) desc")
    for r in c.execute(q):
        print("%-90s %6d %12.1f %10.2f %10.2f %10.2f %8d %8d %7d" % (r[0][:90], r[1], r[2] / 1e3, r[3] / 1e3,
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


def traffic(directory, workload, match="k_solve", calibration=None):
    """One entry of profiles/hbm_traffic.json from the PMC passes in `directory`: HBM bytes (FETCH_SIZE +
    WRITE_SIZE, KB units; raw counters -- K1's loads are 8-byte and dword accesses, so the gfx950 x2
    FETCH_SIZE correction for 16 B/lane streams is not applied) and VALU instructions per launch, stamped
    with the sha of the device sources they were measured on (bench.py reports them only while it matches)."""
    import json
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    vals, name = {}, None
    clocks = []
    for f in sorted(glob.glob(directory + "/*_counter_collection.csv")):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if match in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
                name = r["Kernel_Name"]
                if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
                    # the engine clock under THIS load: busy cycles per XCD (the counter sums the eight XCDs) over the
                    # dispatch's own duration in the same pass
                    ns = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
                    if ns > 0:
                        clocks.append(float(r["Counter_Value"]) / 8.0 / ns)
        for k, v in agg.items():
            vals[k] = sum(v) / len(v)
    waves = int(vals.get("SQ_WAVES", 0))
    # plain float32 fma / add / mul issue at twice the rate of everything else on gfx950 (tools/mb_valu_rates.hip):
    # their count, when the type-mix pass ran, lets bench.py price the instruction stream in cycles
    fast = [vals.get("SQ_INSTS_VALU_%s_F32" % k) for k in ("FMA", "ADD", "MUL")]
    mix = {k: vals[k] for k in sorted(vals) if k.startswith("SQ_INSTS_VALU_")}
    cal = json.load(open(calibration)) if calibration else None
    extra = {}
    if clocks:
        clocks.sort()
        extra["clock_ghz_measured"] = clocks[len(clocks) // 2]
        extra["clock_note"] = ("median over %d k_solve dispatches of GRBM_GUI_ACTIVE / 8 XCDs / (End - Start) of the same "
                               "PMC pass (min %.3f, max %.3f GHz)" % (len(clocks), clocks[0], clocks[-1]))
    if cal:
        extra["hbm_bytes_calibrated"] = (vals["FETCH_SIZE"] * 1024 / cal["fetch_ratio_records"] +
                                         vals["WRITE_SIZE"] * 1024 / cal["write_ratio"])
        extra["calibration_note"] = ("FETCH_SIZE / %.3f + WRITE_SIZE / %.3f: the counters' ratios to the bytes K1's own access "
                                     "shapes move (tools/mb_k1_traffic.hip, %s)" % (cal["fetch_ratio_records"], cal["write_ratio"],
                                                                                      os.path.relpath(calibration)))
    print(json.dumps({workload: {
        **extra,
        "hbm_bytes": (vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024, "valu_insts": vals.get("SQ_INSTS_VALU"),
        "valu_f32_fma_add_mul": sum(fast) if all(v is not None for v in fast) else None, "valu_mix": mix or None,
        "batch": waves, "source_sha": bench.source_sha(),
        "note": "(FETCH_SIZE %.1f KB + WRITE_SIZE %.1f KB) * 1024 per %s launch (%d instances), rocprofv3 --pmc in "
                "separate passes (tools/profile_all.sh), raw counters" % (vals["FETCH_SIZE"], vals["WRITE_SIZE"], name, waves)}}))


if __name__ == "__main__":
    {"kernels": kernels, "pmc": pmc, "traffic": traffic}[sys.argv[1]](*sys.argv[2:])
