#!/usr/bin/env python3
"""Static spill traffic of one K1 variant: SGPR spills (v_writelane/v_readlane through the spill VGPRs)
and VGPR spills (scratch_store/scratch_load), in total and inside the solver loop's source lines.
usage: spill_report.py <mangled-substring> (needs /tmp/asm_profile.s from tools/asm_profile.py)"""
import collections, re, subprocess, sys, os
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
key = sys.argv[1]
subprocess.run([sys.executable, os.path.join(root, "tools/asm_profile.py"), key, "0", "0"], check=True, stdout=subprocess.DEVNULL)
text = open("/tmp/asm_profile.s").read().split("\n")
start = next(i for i, l in enumerate(text) if l.startswith("_Z") and key in l and ":" in l)
end = next(i for i in range(start, len(text)) if ".amdhsa_kernel" in text[i])
src = open(os.path.join(root, "neo_mpc_planner2_amd/csrc/neo_mpc_kernels.hip")).read().split("\n")
lo = next(i for i, l in enumerate(src, 1) if "for (; it < p.max_it && !scan_only; ++it)" in l)
hi = next(i for i, l in enumerate(src, 1) if i > lo and "a search that has ENDED" in l)
body = text[start:end]
spill_regs = collections.Counter(re.search(r"v_writelane_b32 (v\d+),", l).group(1) for l in body if "v_writelane_b32" in l)
regs = set(spill_regs)
cur = 0
c = collections.Counter()
for l in body:
    m = re.match(r"\s*\.loc\s+(\d+)\s+(\d+)\s+\d+", l)
    if m:
        chain = re.findall(r"neo_mpc_kernels\.hip:(\d+):\d+", l)
        cur = (int(chain[-2]) if len(chain) >= 2 else int(chain[-1])) if chain else (int(m.group(2)) if int(m.group(1)) <= 1 else cur)
        continue
    t = l.strip()
    where = "loop" if lo <= cur <= hi else "outside"
    if t.startswith("v_writelane_b32"): c["sgpr_spill_store", where] += 1
    m = re.match(r"v_readlane_b32 s\d+, (v\d+),", t)
    if m and m.group(1) in regs: c["sgpr_spill_reload", where] += 1
    if t.startswith("scratch_store"): c["vgpr_spill_store", where] += 1
    if t.startswith("scratch_load"): c["vgpr_spill_reload", where] += 1
    if t.startswith("v_"): c["valu", where] += 1
    if t.startswith("ds_"): c["lds", where] += 1
for k in ("valu", "lds", "sgpr_spill_store", "sgpr_spill_reload", "vgpr_spill_store", "vgpr_spill_reload"):
    print("%-18s loop %5d   outside %5d" % (k, c[k, "loop"], c[k, "outside"]))
