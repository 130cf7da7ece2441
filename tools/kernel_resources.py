#!/usr/bin/env python3
"""Register / spill / scratch table of every k_solve variant (hipcc -Rpass-analysis=kernel-resource-usage).
usage: kernel_resources.py [substring]"""
import os, re, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "neo_mpc_planner2_amd", "csrc")
key = sys.argv[1] if len(sys.argv) > 1 else "k_solve"
rows = []
for f in ("neo_mpc_kernels.hip", "neo_mpc_riccati.hip"):
    out = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-slp-vectorize", "-Wno-pass-failed", "-I" + os.path.join(root, "include"), "-c",
                          os.path.join(src, f), "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True, cwd=src)
    cur = None
    for line in out.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m: cur = {"name": m.group(1)}; rows.append(cur); continue
        m = re.search(r"remark:\s+([A-Za-z][A-Za-z /\[\]]*?):\s+(\d+)", line)
        if m and cur is not None: cur[m.group(1).strip()] = int(m.group(2))
for r in rows:
    if key not in r["name"]: continue
    t = re.search(r"k_solveI(.*?)EEv", r["name"])
    print("%-34s VGPRs %3d  AGPRs %3d  SGPR spill %3d  VGPR spill %3d  scratch %3d B/lane  occupancy %d" % (
        t.group(1) if t else r["name"][:34], r.get("VGPRs", -1), r.get("AGPRs", -1), r.get("SGPRs Spill", -1), r.get("VGPRs Spill", -1),
        r.get("ScratchSize [bytes/lane]", -1), r.get("Occupancy [waves/SIMD]", -1)))
