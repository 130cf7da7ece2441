#!/usr/bin/env python3
"""Static opcode histogram of one K1 variant, split into the solver loop and the rest (round-3 verdict item 2: "commit a
static opcode histogram of the solver loop of k_solve<4,3,1,true,1024> next to the PMC type mix; name the top
non-arithmetic opcodes").

usage: opcode_histogram.py [kernel-mangled-substring] [--top N] [--by-line] [--branch dense|stagewise]
(K1's solver loop lives in solve_search, inlined into the kernels: an instruction is attributed to its line THERE -- the
frame under the kernel's own call site; --branch keeps one branch of k_solve_routed, whose two branches inline it twice)
Compiles the translation unit with -gline-tables-only -S (device only), attributes every instruction to the kernel's own
source line through the .loc inlining chain (like tools/asm_profile.py) and counts opcodes for the lines of the solver loop
(`for (it = 0; ...` to the exit block).  Static counts: a straight-line count of the code, not of what a wave executes."""
import collections, os, re, subprocess, sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = [a for a in sys.argv[1:] if not a.startswith("--")]
key = args[0] if args else "k_solveILi4ELi3ELi1ELb1ELi1024E"
top = int(sys.argv[sys.argv.index("--top") + 1]) if "--top" in sys.argv else 40
riccati = "ELi0ELi2E" in key or "k_solve_routed" in key
csrc = os.path.join(root, "neo_mpc_planner2_amd/csrc")
src = os.path.join(csrc, "neo_mpc_riccati.hip" if riccati else "neo_mpc_kernels.hip")
out = "/tmp/opcode_histogram.s"
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-Wno-pass-failed"] +
               (["-fno-slp-vectorize", "-mllvm", "-disable-machine-licm"] if riccati else []) +
               ["-gline-tables-only", "-x", "hip", "--cuda-device-only", "-S", src, "-o", out],
               check=True, stderr=subprocess.DEVNULL)
ksrc = open(os.path.join(csrc, "neo_mpc_kernels.hip")).read().split("\n")
loop_lo = next(i + 1 for i, l in enumerate(ksrc) if "for (; it < p.max_it && !scan_only; ++it) {" in l)
branch = sys.argv[sys.argv.index("--branch") + 1] if "--branch" in sys.argv else None
# (the two call sites of solve_search in k_solve_routed: direction 2 = stage-wise, 1 = dense)
site = {"stagewise": next(i + 1 for i, l in enumerate(ksrc) if "solve_search<kMinWavesPerSimd, kSteps, 2, kTame" in l),
        "dense": next(i + 1 for i, l in enumerate(ksrc) if "solve_search<kMinWavesPerSimd, kSteps, 1, kTame" in l)}
# (the loop ends where the exit-hop block of the dense direction, or the epilogue, begins)
loop_hi = next(i for i, l in enumerate(ksrc) if ("a search that has ENDED" in l or "NEO_SEGMENT(1);" in l or "phase 2, second-order directions" in l) and i + 1 > loop_lo)
text = open(out).read().split("\n")
start = next(i for i, l in enumerate(text) if l.startswith("_Z") and key in l and ":" in l)
end = next(i for i in range(start, len(text)) if ".amdhsa_kernel" in text[i])


def klass(op):
    if op.startswith("s_"):
        return "salu/wait" if not op.startswith(("s_waitcnt", "s_nop", "s_barrier")) else op.split("_b")[0]
    if op.startswith(("ds_", "global_", "scratch_", "buffer_", "flat_")):
        return op.split("_")[0] + " (memory)"
    if re.match(r"v_(fma|mul|add|sub|fmac|mac|mad)_f64|v_(fma|mul|add)_f64", op) or op in ("v_fma_f64", "v_mul_f64", "v_add_f64"):
        return "f64 arithmetic"
    if re.match(r"v_(rcp|rsq|sqrt|trig_preop|ldexp|frexp|fract|floor|rndne|ceil|trunc|div).*_f64", op):
        return "f64 special"
    if re.match(r"v_(max|min)_f64", op):
        return "f64 min/max"
    if re.match(r"v_cmp.*_f64|v_cmpx.*_f64|v_cmp_class_f64", op):
        return "f64 compare"
    if re.match(r"v_cvt", op):
        return "convert"
    if re.match(r"v_(fma|mul|add|sub|fmac|mac|mad|fmaak|fmamk|pk_fma|pk_mul|pk_add)_f32|v_pk_(fma|mul|add)_f32", op):
        return "f32 arithmetic"
    if re.match(r"v_(rcp|rsq|sqrt|exp|log|sin|cos).*_f32", op):
        return "f32 special"
    if re.match(r"v_(max|min)", op):
        return "other min/max"
    if re.match(r"v_cmp|v_cmpx", op):
        return "other compare"
    if op.startswith("v_cndmask"):
        return "select (v_cndmask)"
    if op.startswith(("v_mov", "v_accvgpr", "v_swap")):
        return "move (v_mov / accvgpr)"
    if op.startswith(("v_readlane", "v_readfirstlane", "v_writelane")):
        return "lane read/write"
    if "dpp" in op:
        return "dpp"
    if re.match(r"v_(and|or|xor|not|lshl|lshr|ashr|bfe|bfi|perm|alignbit|lshlrev|lshrrev|ashrrev|and_or|or3|xad|lshl_or|lshl_add)", op):
        return "integer logic/shift"
    if re.match(r"v_(add|sub|mul|mad|addc|subb|subrev).*_(u32|i32|u64|i64|co)", op) or op.startswith(("v_add_u32", "v_sub_u32", "v_mul_lo", "v_mul_hi", "v_mad_u", "v_mad_i", "v_add_co", "v_addc", "v_add3")):
        return "integer arithmetic"
    return "other vector"


by_line_prefix = sys.argv[sys.argv.index("--by-line") + 1].split(",") if "--by-line" in sys.argv else None
line_hits = collections.defaultdict(collections.Counter)
cur = 0
skip = False
ops_loop, ops_rest = collections.Counter(), collections.Counter()
cls_loop, cls_rest = collections.Counter(), collections.Counter()
dpp_loop = 0
for l in text[start:end]:
    m = re.match(r"\s*\.loc\s+(\d+)\s+(\d+)\s+\d+", l)
    if m:
        chain = [int(x) for x in re.findall(r"neo_mpc_kernels\.hip:(\d+):\d+", l)]
        own = int(m.group(2)) if int(m.group(1)) <= 1 else None    # (the .loc's own line when it is in neo_mpc_kernels.hip)
        frames = ([own] if own is not None and not chain else []) + chain   # innermost ... outermost (the kernel's call site)
        if own is not None and chain:
            frames = [own] + chain
        skip = branch is not None and site[branch] not in frames
        # the line in solve_search: the frame under the kernel-level call site (a plain kernel: its own line)
        cur = frames[-2] if len(frames) >= 2 else (frames[-1] if frames else cur)
        continue
    t = l.strip()
    if not t or t[0] in ".;_" or t.endswith(":") or skip:
        continue
    op = t.split()[0]
    if not op.startswith(("v_", "s_", "ds_", "global_", "scratch_", "buffer_", "flat_")):
        continue
    name = op + ("_dpp" if ("row_shr" in t or "row_bcast" in t or "quad_perm" in t or "row_shl" in t) and "dpp" not in op else "")
    inside = loop_lo <= cur <= loop_hi
    (ops_loop if inside else ops_rest)[name] += 1
    (cls_loop if inside else cls_rest)[klass(name)] += 1
    if by_line_prefix and inside and name.startswith(tuple(by_line_prefix)):
        line_hits[cur][name] += 1

nl, nr = sum(ops_loop.values()), sum(ops_rest.values())
print("kernel %s: %d instructions in the solver loop (source lines %d-%d), %d outside" % (key, nl, loop_lo, loop_hi, nr))
vl = sum(v for k, v in ops_loop.items() if k.startswith("v_"))
print("solver loop: %d vector, %d scalar, %d memory" % (vl, sum(v for k, v in ops_loop.items() if k.startswith("s_")), nl - vl - sum(v for k, v in ops_loop.items() if k.startswith("s_"))))
print("\n-- solver loop by class")
for k, v in cls_loop.most_common():
    print("%6d  %5.1f %%  %s" % (v, 100.0 * v / nl, k))
print("\n-- solver loop, top %d opcodes" % top)
for k, v in ops_loop.most_common(top):
    print("%6d  %5.1f %%  %s" % (v, 100.0 * v / nl, k))

if by_line_prefix:
    print("\n-- %s by source line of the kernel (solver loop)" % ",".join(by_line_prefix))
    for k in sorted(line_hits, key=lambda k: -sum(line_hits[k].values())):
        print("%5d  line %4d  %s   | %s" % (sum(line_hits[k].values()), k, dict(line_hits[k]), ksrc[k - 1].strip()[:110]))
