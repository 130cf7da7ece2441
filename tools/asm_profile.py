#!/usr/bin/env python3
"""Static instruction count of one kernel per top-level source line (inlined code is attributed to
the call site inside the kernel).  usage: asm_profile.py <kernel-mangled-substring> [lo hi]
Compiles neo_mpc_kernels.hip with -gline-tables-only -S (device only) and parses the .loc chain."""
import collections, os, re, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
key = sys.argv[1]
# the Riccati variants of K1 (k_solve<*, 0, 2, *>: "...ELi0ELi2E...") live in neo_mpc_riccati.hip, built without SLP
riccati = "ELi0ELi2E" in key or "k_solve_routed" in key
src = os.path.join(root, "neo_mpc_planner2_amd/csrc", "neo_mpc_riccati.hip" if riccati else "neo_mpc_kernels.hip")
out = "/tmp/asm_profile.s"
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-Wno-pass-failed"] +
               (["-fno-slp-vectorize", "-mllvm", "-disable-machine-licm"] if riccati else []) +
               ["-gline-tables-only", "-x", "hip", "--cuda-device-only", "-S", src, "-o", out],
               check=True, stderr=subprocess.DEVNULL)
lo, hi = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (0, 10**9)
text = open(out).read().split("\n")
start = next(i for i, l in enumerate(text) if l.startswith("_Z") and key in l and ":" in l)
end = next(i for i in range(start, len(text)) if ".amdhsa_kernel" in text[i])
cur = inner = 0
by_inner = collections.defaultdict(collections.Counter)
cnt = collections.defaultdict(collections.Counter)
for l in text[start:end]:
    m = re.match(r"\s*\.loc\s+(\d+)\s+(\d+)\s+\d+", l)
    if m:
        chain = re.findall(r"neo_mpc_kernels\.hip:(\d+):\d+", l)
        if chain:
            cur = int(chain[-2]) if len(chain) >= 2 else int(chain[-1])   # the frame under the kernel's call site: the line in solve_search (K1 is split into solve_setup / solve_search / solve_finish)
            inner = int(chain[0]) if "neo_mpc_kernels.hip:%s:" % chain[0] in l.split("@[")[0] else -int(m.group(2))
        elif int(m.group(1)) <= 1:
            cur = int(m.group(2))       # not inlined: the kernel's own line
            inner = cur
        continue
    t = l.strip()
    if not t or t[0] in ".;_" or t.endswith(":"):
        continue
    op = t.split()[0]
    if op.startswith(("v_", "s_", "ds_", "global_", "scratch_", "buffer_")):
        kind = "v64" if "f64" in op else op.split("_")[0]
        cnt[cur][kind] += 1
        if lo <= cur <= hi and len(sys.argv) > 4:
            by_inner[inner][kind] += 1
tot = collections.Counter()
for k in sorted(cnt):
    if lo <= k <= hi:
        print(k, dict(cnt[k]))
    tot.update(cnt[k])
print("total", dict(tot))
if by_inner:
    print("-- by innermost source line (negative: a line of a system header) inside [%d, %d]" % (lo, hi))
    for k in sorted(by_inner):
        print(k, sum(by_inner[k].values()), dict(by_inner[k]))
