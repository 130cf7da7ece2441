#!/usr/bin/env python3
"""Study: per-wave start/end times of one C2 launch.  Needs the timing build of the library, which
writes wall_clock64 stamps and HW_ID into entries 6-8 of `solution`:
make -C neo_mpc_planner2_amd/csrc timing; NEO_MPC_LIB=neo_mpc_planner2_amd/libneo_mpc_timing.so python tools/wave_timeline.py"""
import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from neo_mpc_planner2_amd import synthetic
from neo_mpc_planner2_amd.solver import BatchSolver
from neo_mpc_planner2_amd.mpc_optimization_server import README_PARAMS
cfg, cmap, probs, st, warm = synthetic.make_workload("C2", seed=0)
params = dict(README_PARAMS); params.update(control_steps=3)
warm_ticks = int(sys.argv[sys.argv.index("--warm") + 1]) if "--warm" in sys.argv else 0   # the timeline of the T-th warm tick of the closed loop
with BatchSolver(params) as s:
    s.set_costmap(*cmap)
    if not warm_ticks:
        for rep in range(3):
            st, warm = synthetic.make_states(probs, 3)
            cmds, x = s.solve(probs, st, warm)
    else:   # (the fleet loop of neo_mpc_planner2_amd/fleet.py on host arrays: robots moved by their own commands)
        def yaw_of(q):
            return np.arctan2(2 * (q[:, 3] * q[:, 2] + q[:, 0] * q[:, 1]), 1 - 2 * (q[:, 1] ** 2 + q[:, 2] ** 2))
        probs = probs.copy()
        st, warm = synthetic.make_states(probs, 3)
        yaw, pos = yaw_of(probs["cur_q"]).copy(), probs["cur_xy"].copy()
        c_, s_ = np.cos(yaw), np.sin(yaw)
        off = np.stack([c_ * probs["carrot_xy"][:, 0] - s_ * probs["carrot_xy"][:, 1], s_ * probs["carrot_xy"][:, 0] + c_ * probs["carrot_xy"][:, 1]], 1)
        cyw = yaw + yaw_of(probs["carrot_q"])
        probs["control_interval"] = 1 / 30.0
        probs["delta_t"] = 1 / 30.0
        for t in range(warm_ticks + 1):
            cmds, x = s.solve(probs, st, warm)
            v = cmds["vel"]
            yaw = yaw + v[:, 2] / 30.0
            c_, s_ = np.cos(yaw), np.sin(yaw)
            pos = pos + np.stack([c_ * v[:, 0] - s_ * v[:, 1], s_ * v[:, 0] + c_ * v[:, 1]], 1) / 30.0
            probs["cur_xy"], probs["cur_q"] = pos, synthetic.yaw_quat(yaw)
            probs["carrot_xy"][:, 0] = c_ * off[:, 0] + s_ * off[:, 1]
            probs["carrot_xy"][:, 1] = -s_ * off[:, 0] + c_ * off[:, 1]
            probs["carrot_q"] = synthetic.yaw_quat(cyw - yaw)
            probs["cur_vel"] = v
        print("warm tick %d of the closed loop: %d robots stopped by the latch" % (warm_ticks, int((cmds["flags"] & 2 != 0).sum())))
t0, t1, hw = x[:, 6], x[:, 7], x[:, 8].astype(np.int64)
xcc = hw >> 32
hw = hw & 0xffffffff
base = t0.min()
us = lambda t: (t - base) / 100.0     # wall_clock64: 100 MHz
it = cmds["iterations"]
dur = us(t1) - us(t0)
print("launch span %.1f us; wave start spread: p50 %.1f p99 %.1f max %.1f us" % (us(t1).max(), np.median(us(t0)), np.quantile(us(t0), .99), us(t0).max()))
print("wave duration: mean %.1f p50 %.1f p90 %.1f p99 %.1f max %.1f us" % (dur.mean(), np.median(dur), np.quantile(dur, .9), np.quantile(dur, .99), dur.max()))
for k in sorted(set(it)):
    m = it == k
    print("  iterations %2d: n %4d  duration mean %.1f  max %.1f  end max %.1f us" % (k, m.sum(), dur[m].mean(), dur[m].max(), us(t1)[m].max()))
if os.environ.get("NEO_MPC_SEGMENTS"):   # library built with -DNEO_MPC_SEGMENT_TIMING (make segments)
    l0, l1 = x[:, 4], x[:, 5]
    pro, loop, epi = us(l0) - us(t0), us(l1) - us(l0), us(t1) - us(l1)
    print("set-up (records, yaw, reach tile): mean %.1f p50 %.1f p99 %.1f us; iterations: mean %.1f us (%.2f us per iteration); K2 + write-back: mean %.1f p99 %.1f us"
          % (pro.mean(), np.median(pro), np.quantile(pro, .99), loop.mean(), (loop / it).mean(), epi.mean(), np.quantile(epi, .99)))
    print("last set-up ends at %.1f us; first K2 starts at %.1f us" % (us(l0).max(), us(l1).min()))
order = np.argsort(-us(t1))[:12]
seg = bool(os.environ.get("NEO_MPC_SEGMENTS"))
scan_us = x[:, 3] / 100.0 if seg else np.zeros(len(it))
scan_at = np.where(x[:, 2] > 0, us(x[:, 2]), np.nan) if seg else np.zeros(len(it))
if seg:
    sc = scan_us > 0
    print("cell scan: %d of %d waves ran it; duration mean %.1f p50 %.1f p99 %.1f max %.1f us; starts at %.1f .. %.1f us"
          % (sc.sum(), len(it), scan_us[sc].mean(), np.median(scan_us[sc]), np.quantile(scan_us[sc], .99), scan_us[sc].max(), np.nanmin(scan_at), np.nanmax(scan_at)))
print("the waves that end last: (instance, iterations, start us, duration us, end us, hw id, scan us, scan at)")
for i in order:
    print("   %5d  it %2d  start %6.1f  dur %6.1f  end %6.1f  hw %#x  scan %5.1f at %5.1f" % (i, it[i], us(t0)[i], dur[i], us(t1)[i], hw[i], scan_us[i], scan_at[i]))
end = np.sort(us(t1))
print("waves still running at t = 40/60/80/90/100 us:", [(end > t).sum() for t in (40, 60, 80, 90, 100)])
# HW_ID: wave_id[3:0] simd[5:4] pipe[7:6] cu[11:8] sh[12] se[15:13] ... (gfx9 layout)
simd = (hw >> 4) & 3; cu = (hw >> 8) & 15; sh = (hw >> 12) & 1; se = (hw >> 13) & 7
key = (((xcc * 8 + se) * 2 + sh) * 16 + cu) * 4 + simd
u, inv, cnt = np.unique(key, return_inverse=True, return_counts=True)
print("distinct (xcc,se,sh,cu,simd) keys %d; waves per key: min %d max %d" % (len(u), cnt.min(), cnt.max()))
# per SIMD: the sum of its waves' iterations against the time its last wave ends
sums = np.bincount(inv, weights=it); ends = np.array([us(t1)[inv == k].max() for k in range(len(u))])
A = np.stack([sums, np.ones(len(u))], 1); coef = np.linalg.lstsq(A, ends, rcond=None)[0]
print("SIMD end time ~ %.2f us x (sum of iterations on the SIMD) + %.1f us; residual std %.1f us; iteration sums: mean %.1f max %d; the SIMD that ends last: sum %d, %d waves"
      % (coef[0], coef[1], (ends - A @ coef).std(), sums.mean(), sums.max(), sums[np.argmax(ends)], cnt[np.argmax(ends)]))
blk = np.arange(len(it))
print("instance -> SIMD: instances sharing the SIMD of instance 0:", blk[inv == inv[0]][:8], " of instance 1:", blk[inv == inv[1]][:8])
