// neo_mpc_kernels.hip -- gfx950 (CDNA4 / MI355X) kernels of the batched MPC solver.
//
// Replaces, for thousands of independent instances per launch, what the reference does
// one request at a time in Python: `MpcOptimizationServer.optimizer`
// (neo_mpc_planner2/mpc_optimization_server.py:349-403), i.e. the SciPy SLSQP `minimize`
// call (py:363-364) over `objective` (py:204-269) with the box bounds and the
// translational-speed disc (py:125-134, 157-158), followed by the low-pass, collision
// check, stop latch, acceleration clamp and warm-start shift (py:365-403).
//
//   K1 k_solve        one 64-lane wavefront per instance.  Per iteration the wave
//                     evaluates 64 candidate control sequences at once (one rollout per
//                     lane): lanes 0-31 walk the projected proximal-gradient arc at 32
//                     step sizes, lanes 32-63 a projected second-order direction at 32 step
//                     lengths -- dense Newton at control_steps 3 (finite-difference Hessian of the
//                     analytic gradient, one column per lane, solved in registers; the headline
//                     specialisation), the stage-wise (Riccati) Gauss-Newton sweep of riccati.h at
//                     every other control_steps (rollout/adjoint as DPP prefix scans, lane = stage;
//                     damped beyond 8 steps; wall model for costmap steps; in free space the full
//                     step is tried alone before the search), projected L-BFGS on request; the lowest
//                     objective wins (wave arg-min).  Iterates, gradients and the direction's
//                     state live in LDS; the (2R+1)^2 costmap reach tile is staged into LDS
//                     once per solve with coalesced dword loads.  float64 throughout (the arc
//                     search compares objective values, which resolves the minimiser to
//                     sqrt(eps); MI355X has full-rate vector f64); the Newton systems themselves
//                     are float32 (they only yield a direction).
//   K2 postprocess    py:365-403, fused as the epilogue of K1 and launchable on its own.
//   K3 k_ingest       raw nav2 costmap -> device map with a lethal border and 128-byte
//                     row pitch (16 B per lane, HBM-streaming).
//   K4 k_carrot       the step before the solver: plan pruning + look-ahead point
//                     (src/NeoMpcPlanner.cpp:83-104, 157-189, 221-232), HBM-streaming.
//   k_objective       py:204-269 for given controls (parity checks of the objective).
//
// No MFMA: a 3*control_steps-variable problem has no dense contraction.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

#include "neo_mpc_device.h"
#include "wave_ops.h"
#include "fast_math.h"
#include "solver_context.h"
#include "solver_rules.h"
#include "costmap.h"
#include "feasible_set.h"
#include "rollout.h"
#include "riccati.h"
#include "adjoint.h"
#include "tangent_cone.h"
#include "dense_newton.h"
#include "lbfgs.h"
#include "cell_scan.h"

namespace neo_mpc {
namespace {

// the stop-rule constants: solver_rules.h (shared with the host side and with the CPU mirror)
constexpr int kStallIterations = NEO_RULE_STALL_ITERATIONS;
constexpr int kBlockedRun = NEO_RULE_BLOCKED_RUN;
constexpr double kBlockedStep = NEO_RULE_BLOCKED_STEP;
constexpr double kFinalFracGaussNewton = NEO_RULE_FINAL_FRAC_GN;
constexpr double kWindowStep = NEO_RULE_WINDOW_STEP;
constexpr int kClosingRun = NEO_RULE_CLOSING_RUN;
constexpr int kLateIteration = NEO_RULE_LATE_ITERATION;

// Study build (make timing -> libneo_mpc_timing.so): shader-clock stamps at the phase boundaries of
// solver iteration 2 in entries 0-5 of `solution` (tools/phase_timing.py); wall-clock start and end of
// the wave and its HW_ID in entries 6-8 (tools/wave_timeline.py; control_steps >= 3).
#ifdef NEO_MPC_PHASE_TIMING
// (-DNEO_MPC_SEGMENT_TIMING on top: wall-clock stamps at the first and behind the last solver iteration replace
// the phase clocks of entries 4-5 -- set-up, iterations and K2 of every wave, tools/wave_timeline.py)
#ifdef NEO_MPC_SEGMENT_TIMING
#define NEO_SEGMENT_DECL unsigned long long seg_t0 = 0, seg_t1 = 0, seg_scan = 0, seg_scan_at = 0
#define NEO_SEGMENT(k) seg_t##k = wall_clock64()
#define NEO_SEGMENT_SCAN_BEGIN() seg_scan_at = wall_clock64(); seg_scan = seg_scan_at
#define NEO_SEGMENT_SCAN_END() seg_scan = wall_clock64() - seg_scan
#define NEO_SEGMENT_DUMP()                                                                                   \
  {                                                                                                          \
    SolveArgs ad;                                                                                            \
    fresh_args<kSteps, kStaticTile, kLayoutSteps, kRouted>(ad);                                              \
    const int nvd = 3 * ad.p.n;                                                                              \
    if (ad.solution && lane_again() == 0 && nvd >= 9) {                                                      \
      ad.solution[(size_t)b * nvd + 4] = (double)seg_t0; ad.solution[(size_t)b * nvd + 5] = (double)seg_t1;  \
      ad.solution[(size_t)b * nvd + 3] = (double)seg_scan; ad.solution[(size_t)b * nvd + 2] = (double)seg_scan_at; \
    }                                                                                                        \
  }
#else
#define NEO_SEGMENT_DECL
#define NEO_SEGMENT(k)
#define NEO_SEGMENT_SCAN_BEGIN()
#define NEO_SEGMENT_SCAN_END()
#define NEO_SEGMENT_DUMP()
#endif
#define NEO_WAVE_START const unsigned long long wave_t0 = wall_clock64()
#define NEO_WAVE_END_ARGS(args)                                                                     \
  if ((args).solution && lane == 0 && (args).p.n >= 3) {                                            \
    const int nvw = 3 * (args).p.n;                                                                \
    unsigned int hw_id, xcc_id;                                                                    \
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw_id));                            \
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc_id));                          \
    (args).solution[(size_t)b * nvw + 6] = (double)wave_t0;                                        \
    (args).solution[(size_t)b * nvw + 7] = (double)wall_clock64();                                 \
    (args).solution[(size_t)b * nvw + 8] = (double)hw_id + 4294967296.0 * (double)(xcc_id & 15u);  \
  }
#define NEO_PHASE_DECL long long phase_clock[8]
#define NEO_PHASE(k) phase_clock[k] = clock64()
#define NEO_PHASE_DUMP()                                                                            \
  if (it == 2 && a.solution && lane == 0)                                                          \
    for (int k = 0; k < 6; ++k) a.solution[(size_t)b * nv + k] = (double)(phase_clock[k + 1] - phase_clock[k])
#else
#define NEO_WAVE_START
#define NEO_WAVE_END_ARGS(args)
#define NEO_SEGMENT_DECL
#define NEO_SEGMENT(k)
#define NEO_SEGMENT_SCAN_BEGIN()
#define NEO_SEGMENT_SCAN_END()
#define NEO_SEGMENT_DUMP()
#define NEO_PHASE_DECL
#define NEO_PHASE(k)
#define NEO_PHASE_DUMP()
#endif
// largest control_steps the run-time-sized Newton kernel takes (a 24 x 24 system: rows in registers)
constexpr int kNewtonMaxSteps = 8;

// ---------------------------------------------------------------- K1: phases
// K1 runs in phases -- set-up, search, cell scan, K2 -- and the search can be taken up again behind the scan.  Each phase
// starts from a FRESH copy of the launch arguments (re-read from the kernarg segment through a pointer the compiler cannot
// see through) and of the per-instance constants (re-read from the tolerance block of LDS, where the set-up leaves them): the
// vector-register residents of the solver loop are then dead outside it.  Allocated as ONE live range across the scan they
// came out spilled -- and reloaded from scratch inside the loop, +20 % on a C2 launch for code that runs once per solve.
// kRouted: the stage-wise branch of the routed control_steps-3 kernel (k_solve_routed).  Its LDS carve-up is the stage-wise
// one for three stages laid INSIDE the dense kernel's (records, tolerance block, term table, iterate and gradients sit at the
// same offsets in both; the reach tile stays where the set-up staged it), control_steps is the constant 3 -- the run-time-sized
// code below folds to three stages -- and the prox-only zone around the kink is the stage-wise direction's.
constexpr int kRoutedSteps = 3;
constexpr LdsLayout make_routed_stagewise_layout() {
  LdsLayout l = make_lds_layout(kRoutedSteps, 0, true, true);
  constexpr LdsLayout dense = make_lds_layout(kRoutedSteps, 4, false);
  l.tile = dense.tile;
  l.total_bytes = dense.total_bytes;
  return l;
}
static_assert(make_lds_layout(kRoutedSteps, 0, true, true).tile <= make_lds_layout(kRoutedSteps, 4, false).tile,
              "the stage-wise arrays of three stages fit in front of the dense kernel's reach tile");
static_assert(make_lds_layout(kRoutedSteps, 0, true, true).gr == make_lds_layout(kRoutedSteps, 4, false).gr,
              "records, iterate and gradients share their offsets");
template <int kSteps, int kStaticTile, int kLayoutSteps, bool kRouted = false>
__device__ __forceinline__ void fresh_args(SolveArgs& a) {
  auto kp = (const __attribute__((address_space(4))) char*)__builtin_amdgcn_kernarg_segment_ptr();   // (k_solve's only argument: offset 0)
  asm volatile("" : "+s"(kp));
  a = *(const SolveArgs*)kp;
  if (kRouted) {
    constexpr LdsLayout kLayout = make_routed_stagewise_layout();
    const int tile_w = a.lds.tile_w, tile_h = a.lds.tile_h, reach = a.lds.reach;
    a.lds = kLayout;
    a.lds.tile_w = tile_w; a.lds.tile_h = tile_h; a.lds.reach = reach;
    a.p.n = kRoutedSteps;
  } else if (kSteps || kStaticTile) {  // compile-time LDS offsets (lbfgs_memory is 4 in the specialisations' layout)
    constexpr LdsLayout kStaticLayout = make_lds_layout(kLayoutSteps, kSteps ? 4 : 0, false);
    const int tile_w = a.lds.tile_w, tile_h = a.lds.tile_h, reach = a.lds.reach;
    a.lds = kStaticLayout;
    a.lds.tile_w = tile_w; a.lds.tile_h = tile_h; a.lds.reach = reach;
    if (kSteps) a.p.n = kSteps;
  }
}
// The parameters the solver loop reads are detached from the wide scalar loads that bring the
// kernel arguments in: a spilled s_load_dwordx16 tuple comes back whole (16 v_readlane) for every
// use of one of its members; as values of their own they are reloaded pair by pair.
__device__ __forceinline__ void own_loop_params(SolveArgs& a) {
  auto own = [](double& v) { asm volatile("" : "+s"(v)); };
  own(a.p.dt); own(a.p.wt_n); own(a.p.wo_n); own(a.p.wc_n); own(a.p.wterm_o); own(a.p.r);
  own(a.p.lo[2]); own(a.p.hi[2]);
  own(a.map.origin_x); own(a.map.origin_y); own(a.map.resolution); own(a.map.inv_resolution);
}
// the per-instance constants as the set-up left them in LDS (request record + tolerance block); wave-uniform: scalar
// registers -- except, kVectorPose, the four the costmap lookup of every stage of every candidate needs (k_solve says why)
template <bool kVectorPose>
__device__ __forceinline__ void ctx_from_lds(const SolveArgs& a, const double* L, Ctx& c) {
  const double* P = L + a.lds.prob;
  const double* t = L + a.lds.tol;
  c.cx = lane_value(P[P_CARROT_X], 0); c.cy = lane_value(P[P_CARROT_Y], 0);
  c.tyaw = lane_value(t[T_TYAW], 0); c.fyaw = lane_value(t[T_FYAW], 0);
  c.v0 = lane_value(P[P_VEL], 0); c.v1 = lane_value(P[P_VEL + 1], 0); c.v2 = lane_value(P[P_VEL + 2], 0);
  c.c0 = t[T_C0]; c.s0 = t[T_S0]; c.X0 = P[P_CUR_X]; c.Y0 = P[P_CUR_Y];
  if (kVectorPose) asm volatile("" : "+v"(c.c0), "+v"(c.s0), "+v"(c.X0), "+v"(c.Y0));
  else { c.c0 = lane_value(c.c0, 0); c.s0 = lane_value(c.s0, 0); c.X0 = lane_value(c.X0, 0); c.Y0 = lane_value(c.Y0, 0); }
  const int* ti = reinterpret_cast<const int*>(t + T_TILE);
  c.tile_x0 = uniform_int(ti[0]); c.tile_y0 = uniform_int(ti[1]); c.tile_geom = uniform_int(ti[2]);
  c.konst = 0.0; c.true_yaw = 0.0;   // (inside the search f excludes the constant terms; K2 reads both from the block)
}

// ---------------------------------------------------------------- K1
// kSteps > 0: specialisation for control_steps == kSteps -- every lane keeps its candidate's controls
// and sin/cos in registers, so the winner is stored without being recomputed and the next adjoint
// sweep needs no trigonometry.  kSteps == 0: any control_steps (LDS-only path).
// kNewton: lanes 32-63 walk the projected Newton direction -- control_steps == kSteps, or with
// kSteps == 0 any control_steps <= kNewtonMaxSteps (measured against L-BFGS on the 1000^2 map,
// 65 536 instances: +28 % at control_steps 1, +54 % at 2, +50 % at 4, +38 % at 5-7, +42 % at 8).
// kTame: instantiation for parameter sets (the README's among them) whose max_vel_trans disc lies
// inside the vx/vy box -- the box/disc corner cases of the projection and of the tangent cone drop
// out -- and whose heading cannot leave [-pi/4, pi/4] within the horizon -- no range reduction in
// the rollout's sin/cos.
// kStaticTile > 0 (with kSteps > 0): LDS is a static array sized for the compile-time layout plus a reach
// tile of at most kStaticTile bytes -- every LDS address is then an instruction immediate instead of a
// "dynamic LDS base + offset" value that lives in (and gets spilled from) a scalar register.
// kDir: search direction of lanes 32-63 -- 0 projected L-BFGS, 1 projected Newton with the dense system
// (above), 2 projected Newton solved stage by stage (riccati.h; kSteps == 0 only, any control_steps).
// What the phases hand to one another in registers (everything else goes through LDS).
struct SolveCarry {
  int flags;                 // NEO_MPC_FLAG_RESET (set-up -> K2)
  bool cold;                 // x0 == 0 (wave-uniform: every lane scans the same LDS values)
  bool wall_in_reach;        // a lethal cell in the reach tile (set-up -> the routed kernel's branch)
  double f = INFINITY;       // objective at u; f(x0) comes out of the first candidate pass: lane 0 evaluates x0 itself there
  int it = 0, nfev = 1, status = NEO_MPC_STATUS_MAX_ITER;
};

// ---- phase 0: set-up.  Records, reset, footprint, reach tile; the per-instance constants and the stop tolerances go to
//      the tolerance block of LDS (layout: solver_context.h), where every later phase reads them; x0.
//      Returns false when the instance makes no request this tick (nothing more to do).
template <int kSteps, bool kTame, int kStaticTile, int kLayoutSteps>
__device__ __forceinline__ bool solve_setup(double* L, uint32_t b, int lane, SolveCarry& s) {
  constexpr bool kCovered = kStaticTile > 0;   // the reach tile is there and covers every lookup (costmap.h cell_raw)
  int flags;
  bool cold = true;
  {
    SolveArgs a;
    fresh_args<kSteps, kStaticTile, kLayoutSteps>(a);
    const DevParams& p = a.p;
    const int n = kSteps ? kSteps : p.n, nv = 3 * n;
    load_records(a, L, b, lane);
    // no request for this robot this tick (the plugin threw before its service call, cpp:234-236; K4 status 3): the node's
    // state does not advance: state record and warm start keep their bytes, the outputs say "skipped" (rollout.h)
    if (uniform_int(reinterpret_cast<const int*>(L + a.lds.prob)[PI_SKIP]) == 1) { skip_instance(a, b, lane); return false; }
    select_map(a.map, L + a.lds.prob);
    flags = reset_and_warm(a, L, b, lane) ? NEO_MPC_FLAG_RESET : 0;
    // (the footprint cost -- the request's own, or the raster's -- waits for K2 in the request record's slot in LDS: as a
    // register it was live across the whole kernel and, at four waves per SIMD, spilled to scratch: the only scratch of the
    // headline kernel, 1.5 KB of memory traffic per solve against 885 B of algorithmic bytes)
    const double fcost = footprint_cost(a, L, b, lane);
    if (lane == 0) L[a.lds.prob + P_FOOTPRINT] = fcost;
    Ctx c;
    make_ctx_wave(p, a.map, L + a.lds.prob, fcost, c, lane);
    load_tile(a, c, L, lane);
    s.wall_in_reach = (c.tile_geom & kTileWall) != 0;
    if (s.wall_in_reach) flags |= NEO_MPC_FLAG_WALL_IN_REACH;
    // The stop tolerances are read once per iteration: from LDS, so that they do not sit in (and get
    // spilled from) scalar registers all through the loop.
    // So do two per-instance constants the loop has no use for: the request's true yaw (K2 only) and
    // the part of the objective that does not depend on u -- inside the loop f excludes it.
    if (lane == 0) {
      double* t = L + a.lds.tol;
      t[T_XTOL] = p.xtol; t[T_EARLY] = p.early_tol; t[T_FINAL] = p.final_tol; t[T_FTOL] = p.ftol;
      t[T_STALL] = p.stall_step; t[T_WTOL] = p.wtol; t[T_WTOL_LATE] = p.wtol_late; t[T_KINK] = p.kink_radius;
      t[T_KONST] = c.konst; t[T_TRUE_YAW] = c.true_yaw;
      t[T_HOP_DROP] = p.hop_min_drop; t[T_HOP_RANGE] = p.hop_range;
      t[T_BTOL_MAP] = p.btol_map; t[T_BTOL_FREE] = p.btol_free;
      reinterpret_cast<int*>(t + T_HOP_STAGE)[kHopLanes] = 0;   // no hop candidates yet
      t[T_C0] = c.c0; t[T_S0] = c.s0; t[T_TYAW] = c.tyaw; t[T_FYAW] = c.fyaw;
      int* ti = reinterpret_cast<int*>(t + T_TILE);
      ti[0] = c.tile_x0; ti[1] = c.tile_y0; ti[2] = c.tile_geom;
    }
    double* u = L + a.lds.u;
    // x0 clipped to the feasible set (SciPy clips x0 to the bounds, _slsqp_py.py:268)
    for (int i = lane; i < n; i += kLanes) project_block<kTame>(p, u[3 * i], u[3 * i + 1], u[3 * i + 2]);
    WAVE_SYNC();
    for (int k = 0; k < nv; ++k) cold = cold && (u[k] == 0.0);
    ctx_from_lds<false>(a, L, c);
    // The warm start is the previous solution shifted by a WHOLE control step (py:198-202: block i <- block i + 1, the
    // FILTERED first control last, py:366-367) although only one control interval -- an eighth of a step at 30 Hz and the
    // README's horizon -- has passed: the previous solution itself, i.e. the shift undone with the first block as the solver
    // left it (kept in the state record by K2: S_PREV_U0; the filtered one where a caller's record does not carry it), is
    // usually much closer to this tick's minimiser -- and IS the minimiser for a robot the collision latch has stopped.
    // In free space (no costmap term under either rollout: one basin) the search starts from whichever of the two has the
    // lower objective (lane 0 rolls out the warm start, lane 1 the un-shifted one).  On the costmap it starts where the
    // reference starts, and the un-shifted point, when it has the lower objective, is ONE CANDIDATE of the first iteration
    // (lane kAltLane, competing by objective value like every other lane: it wins where the problem has not changed, and loses
    // to the first step from the reference's start where that leads into another basin -- starting from it outright ended one
    // recorded call of the reference 1.9e-3 above it).  The warm start handed BACK is the reference's shift as ever (K2).
    // Closed loop of 4096 robots (CPU mirror): 6.2 -> 4.45 -> 3.6 iterations per warm tick, per-tick maximum 13 -> 10.
    {
      double* S = L + a.lds.state;
      int* Si = reinterpret_cast<int*>(S);
      const bool has_prev = uniform_int(Si[SI_HAS_PREV]) == 1 && !(flags & NEO_MPC_FLAG_RESET);
      WAVE_SYNC();
      if (lane == 0) {
        double q0 = has_prev ? S[S_PREV_U0] : u[nv - 3], q1 = has_prev ? S[S_PREV_U0 + 1] : u[nv - 2], q2 = has_prev ? S[S_PREV_U0 + 2] : u[nv - 1];
        project_block<kTame>(p, q0, q1, q2);
        S[S_PREV_U0] = q0; S[S_PREV_U0 + 1] = q1; S[S_PREV_U0 + 2] = q2;
        Si[SI_HAS_PREV] = 0;
      }
      WAVE_SYNC();
    }
    if (!cold && n > 1 && !(p.compat & kCompatNoUnshift) && p.max_it < kDumpGradient) {   // (not in the test hooks: they dump AT the given point)
      const double* A0 = L + a.lds.state + S_PREV_U0;
      double ts = 0.0;
      const double fs = rollout_cost<kSteps, kTame, kCovered>(
          a, c, L,
          [&](int i, double& b0, double& b1, double& b2) {
            const double* src = lane == 1 ? (i == 0 ? A0 : u + 3 * (i - 1)) : u + 3 * i;
            b0 = src[0]; b1 = src[1]; b2 = src[2];
          },
          NoRecord(), &ts);
      const double f_warm = lane_value(fs, 0), f_alt = lane_value(fs, 1);
      const bool free_both = lane_value(ts, 0) == 0.0 && lane_value(ts, 1) == 0.0;
      if (f_alt < f_warm && free_both) {
        double v0 = 0.0, v1 = 0.0, v2 = 0.0;   // (nv <= 192: up to three elements per lane)
        const int k0 = lane, k1 = lane + kLanes, k2 = lane + 2 * kLanes;
        if (k0 < nv) v0 = k0 >= 3 ? u[k0 - 3] : A0[k0];
        if (k1 < nv) v1 = u[k1 - 3];
        if (k2 < nv) v2 = u[k2 - 3];
        WAVE_SYNC();
        if (k0 < nv) u[k0] = v0;
        if (k1 < nv) u[k1] = v1;
        if (k2 < nv) u[k2] = v2;
      } else if (f_alt < f_warm && lane == 0) {
        reinterpret_cast<int*>(L + a.lds.state)[SI_HAS_PREV] = kAltArmed;
      }
      WAVE_SYNC();
    }
  }
  s.flags = flags; s.cold = cold;
  return true;
}

// ---- phases 1 and 2: the search, the cell scan behind it, and the search once more behind a scan that paid.
// kSteps > 0: specialisation for control_steps == kSteps -- every lane keeps its candidate's controls
// and sin/cos in registers, so the winner is stored without being recomputed and the next adjoint
// sweep needs no trigonometry.  kSteps == 0: any control_steps (LDS-only path).
// kDir: search direction of lanes 32-63 -- 0 projected L-BFGS, 1 projected Newton with the dense system: control_steps ==
// kSteps, or with kSteps == 0 any control_steps <= kNewtonMaxSteps (measured against L-BFGS on the 1000^2 map, 65 536
// instances: +28 % at control_steps 1, +54 % at 2, +50 % at 4, +38 % at 5-7, +42 % at 8), 2 projected Newton solved stage by
// stage (riccati.h; kSteps == 0 only, any control_steps).
// kTame: instantiation for parameter sets (the README's among them) whose max_vel_trans disc lies
// inside the vx/vy box -- the box/disc corner cases of the projection and of the tangent cone drop
// out -- and whose heading cannot leave [-pi/4, pi/4] within the horizon -- no range reduction in
// the rollout's sin/cos.
// kStaticTile > 0: LDS is a static array sized for the compile-time layout plus a reach
// tile of at most kStaticTile bytes -- every LDS address is then an instruction immediate instead of a
// "dynamic LDS base + offset" value that lives in (and gets spilled from) a scalar register.
// kRouted: the stage-wise branch of k_solve_routed (fresh_args).
// Returns true when a test hook has dumped what it was asked for (the kernel ends there).
template <int kMinWavesPerSimd, int kSteps, int kDir, bool kTame, int kStaticTile, int kLayoutSteps, bool kRouted = false>
__device__ __forceinline__ bool solve_search(double* L, uint32_t b, SolveCarry& sc) {
  constexpr bool kCovered = kStaticTile > 0;   // the reach tile is there and covers every lookup (costmap.h cell_raw)
  constexpr bool kNewton = kDir == 1;    // dense system in registers
  constexpr bool kRiccati = kDir == 2;   // stage-wise recursion
  constexpr bool kSecond = kDir != 0;    // either: Newton stop rules, no quasi-Newton state
  // the control_steps-3 stop rules of the dense direction -- the window rule on every run of three iterations, the blocked-run
  // rule, closing-in behind two blocked iterations -- also end the stage-wise searches of the routed kernel: what ends a search
  // depends on the horizon, not on how the direction was computed (solver_rules.h neo_rules_derive_routed)
  constexpr bool kBlockedRule = kNewton || kRouted;
  static_assert(!kRiccati || kSteps == 0 || kRouted, "the stage-wise direction: run-time-sized, or the routed kernel's control_steps-3 branch");
  static_assert(!kRouted || (kRiccati && kSteps == kRoutedSteps), "the routed branch is the stage-wise one, three stages in registers");
  // (kSteps > 0 with the stage-wise direction -- the routed branch: candidates in registers, unrolled rollouts, the winner
  // stored without being recomputed like the dense specialisation; gradient and sweep run lane = stage on three lanes)
  constexpr int kFew = (kRiccati && kSteps > 0) ? kSteps : 0;
  // Newton: control_steps == kSteps, or (kSteps == 0) any control_steps <= kNewtonMaxSteps -- the
  // system's arrays are sized for the bound and every loop over them is guarded by the run-time size
  constexpr int kNwSteps = !kNewton ? 1 : kSteps ? kSteps : kNewtonMaxSteps;
  // the four constants the costmap lookup of every stage of every candidate needs (world position = X0 + Rot(psi0) (x, y))
  // stay in VECTOR registers (every lane holds the same value).  The scalar file is over-subscribed -- a hundred
  // scalars are spilled to vector lanes -- and each use of a spilled pair costs two v_readlane and a wait state: twelve
  // lane reads per stage.  (Round 4: the three-address polynomial kernels freed nine vector registers.)
  // (the general kernels have no vector register to spare at four waves per SIMD: scalar there, as before)
  constexpr bool kVectorPose = kTame;
  constexpr int kRegSteps = kSteps ? kSteps : 1;
  constexpr int kPairs = kSteps ? 4 : NEO_MPC_MAX_LBFGS_MEMORY;  // specialisations: lbfgs_memory <= 4
  double f = sc.f;
  const bool cold = sc.cold;
  // ---- what the search carries from one iteration to the next -- and across the cell scan when it is taken up again
  int npairs = 0, head = 0, nfev = sc.nfev, it = sc.it, status = sc.status, stall = 0;
  int blocked_run = 0;   // dense Newton: consecutive iterations not won by a decent Newton step
  int nblocked = 1;      // consecutive iterations not won by a Newton step of at least half its length (or won by a hop)
  double u_term = 0.0;   // dense Newton: sum of the costmap terms under the current iterate's rollout (0: every stage in a free cell)
  double gain1 = INFINITY, gain2 = INFINITY;  // objective decrease of the previous two iterations
  bool final_step = false;
  double alpha = 1.0;
  bool scanned = false;   // the cell scan has had its turn (cell_scan.h)
  int nscans = 0;         // ... scans of its round so far: a scan that found a cheaper cell is followed by another from the new point
  bool scan_only = false; // this pass of the loop below only scans again
  // (the lane index is not kept in a register across the loop -- the stage-wise kernels at four waves per SIMD parked it in
  // scratch and reloaded it at the top of every iteration: it is re-derived from the hardware's lane mask count, seeded
  // with an opaque zero so that the compiler cannot hoist it either)
  auto lane_again = []() {
    int zero;
    asm volatile("v_mov_b32 %0, 0" : "=v"(zero));
    return (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, (unsigned)zero));
  };
  NEO_SEGMENT_DECL;
  NEO_SEGMENT(0);
  for (;;) {   // (the search; taken up again behind a cell scan that paid)
  // ---- phase 1: the search
  SolveArgs a;
  fresh_args<kSteps, kStaticTile, kLayoutSteps, kRouted>(a);
  own_loop_params(a);
  select_map(a.map, L + a.lds.prob);
  const DevParams& p = a.p;
  const int n = kSteps ? kSteps : p.n, nv = 3 * n, mem = p.mem;
  double cand[3 * kRegSteps], cand_sn[kRegSteps], cand_cs[kRegSteps];
  bool have_trig = false;  // ACS/ASN already hold sin/cos of the rollout at u
  Ctx c;
  ctx_from_lds<kVectorPose>(a, L, c);
  double* u = L + a.lds.u;
  double* gs = L + a.lds.gs;
  double* gt = L + a.lds.gt;
  double* gr = L + a.lds.gr;
  double* d = L + a.lds.d;
  double* u_prev = L + a.lds.u_prev;
  double* gt_prev = L + a.lds.gt_prev;
  double* u_new = L + a.lds.u_new;
  double* ACS = L + a.lds.cs;
  double* ASN = L + a.lds.sn;
  int* AMODE = reinterpret_cast<int*>(L + a.lds.mode);  // [4n]: mode, wfroz, near, near_prev
  // Newton: one float32 record per control block (projector + block curvature), written by the
  // tangent-cone pass; it lives in the cs..rt step arrays, which the Newton kernel does not use
  static_assert(2 * 7 >= kNewtonRecord, "Newton records do not fit the step arrays");

  if (!scanned && nscans == 0) {
    const int lane = lane_again();
    for (int i = lane; i < n; i += kLanes) { AMODE[4 * i + 2] = 0; AMODE[4 * i + 3] = 0; }
    // (routed stage-wise branch: this direction's prox-only zone around the kink -- the set-up wrote the dense direction's)
    if (kRouted && lane == 0) L[a.lds.tol + T_KINK] = p.kink_radius_stagewise;
    // Long horizons (Riccati direction): the curvature of a block falls with 1/N^2, so the proximal step starts
    // longer; and the Newton step is long along the valleys in which neighbouring blocks trade displacement and
    // leaves the region where the model holds -- Levenberg-Marquardt damping mu (in units of one stage's tracking
    // weights, riccati_prepare), relaxed x1/4 after an iteration won by the (nearly) full Newton step, tightened
    // x4 after one won by a proximal step or a short Newton step.  Nothing of it at control_steps <= 8.
    alpha = kRiccati ? fmax(1.0, n * 0.125) : 1.0;
    // (routed stage-wise branch: what the search carries between iterations waits in the tolerance block of LDS -- kept in
    // registers across the 64-candidate pass, whose candidates sit in registers themselves, it came out spilled to scratch)
    if (kFew && lane == 0) { double* t = L + a.lds.tol; t[T_GAIN1] = INFINITY; t[T_GAIN2] = INFINITY; t[T_ALPHA] = alpha; }
    // (mu lives in the tolerance block of LDS: two scalar registers fewer across the loop)
    if (kRiccati && lane == 0) L[a.lds.tol + T_MU] = n > 8 ? (double)(n - 8) * 0.125 : 0.0;
  }
  // Riccati: a block may be sent straight onto the kink u_i = v_cur only when v_cur is feasible
  bool v_feasible = false;
  if (kRiccati) {
    double b0 = c.v0, b1 = c.v1, b2 = c.v2;
    asm volatile("" : "+v"(b0), "+v"(b1), "+v"(b2));   // (copies of their own: vector copies of v_cur must not outlive this test)
    project_block<kTame>(p, b0, b1, b2);
    v_feasible = b0 == c.v0 && b1 == c.v1 && b2 == c.v2;
  }
  // this lane's step multiplier: the one lane-derived constant worth two registers for the whole loop
  // (half of the table sits in constant memory: re-reading it would put a global load on every
  // iteration's critical path)
  double my_scale;
  {
    int l0 = lane_again();
    my_scale = kRiccati ? 0.0 : lane_scale<kSecond>(l0);
  }
  for (; it < p.max_it && !scan_only; ++it) {
    // The lane index is re-read opaquely every iteration: otherwise the compiler hoists two dozen
    // lane-derived constants (step multipliers, compare masks, LDS addresses) out of the loop and,
    // at 4 waves/SIMD, parks them in scratch -- recomputing them costs a few integer operations.
    int lane = lane_again();
    // stage-wise direction: this iteration's sweep carries the second-order terms of the rollout step
    const bool exact_step = kRiccati && nblocked == 0;
    // (same trick for the tolerance block: an opaque LDS offset keeps its loads inside the loop and in
    // the LDS address space -- a volatile pointer would turn them into flat loads with a full wait each)
    int tol_off = a.lds.tol;
    asm volatile("" : "+s"(tol_off));
    const double* TOL = L + tol_off;
    NEO_PHASE_DECL;
    NEO_PHASE(0);
    // ---- adjoint gradient of the tracking + terminal cost
    constexpr int kVars = 3 * kNwSteps;  // compile-time bound of the Newton system (= its size when kSteps > 0)
    const int nvr = kSteps ? kVars : nv;  // its size
    bool free_path = false;   // Riccati: every stage of the rollout at u sits in a free cell (raw cost 0)
    int nhops = 0;            // Riccati: hop candidates of this iteration (wave-uniform; the table is in the tolerance block)
    float hc[kVars];     // Newton: column `lane` of the Hessian, then row `lane` (float32, see below)
    float newton_sol = 0.0f;  // Newton: entry `lane` of the direction
    if (kNewton) {
      double hcol[kSteps ? kVars : 1];  // control_steps specialisation: gradient of this lane's perturbed copy
      // Every lane runs the rollout + adjoint sweep on its own copy of u: lane k < 3N perturbs
      // coordinate k by h, the other lanes leave u alone.  One pass therefore yields the gradient
      // (any unperturbed lane) and all 3N Hessian columns by forward differences -- the sweep
      // costs the same whether the lanes agree or not.
      const double hstep = 1e-6, inv_h = 1.0 / hstep;
      // forward: keep only sin/cos per step; the reverse pass rebuilds increments and residuals
      // while it unwinds x, y, theta (fewer live registers -> one more wave per SIMD)
      double pcs[kNwSteps], psn[kNwSteps];
      double x = 0.0, y = 0.0, th = 0.0;
    #pragma unroll
      for (int i = 0; i < kNwSteps; ++i) {
        if (kSteps || i < n) {
          const double vx = u[3 * i] + (lane == 3 * i ? hstep : 0.0);
          const double vy = u[3 * i + 1] + (lane == 3 * i + 1 ? hstep : 0.0);
          const double w = u[3 * i + 2] + (lane == 3 * i + 2 ? hstep : 0.0);
          th += w * p.dt;
          sincos_heading<kTame>(th, &psn[i], &pcs[i]);
          x += (vx * pcs[i] - vy * psn[i]) * p.dt;
          y += (vx * psn[i] + vy * pcs[i]) * p.dt;
        }
      }
      double SX = 0.0, SY = 0.0, ST = 0.0;
    #pragma unroll
      for (int k = kNwSteps - 1; k >= 0; --k) {
        if (kSteps || k < n) {
          const double vx = u[3 * k] + (lane == 3 * k ? hstep : 0.0);
          const double vy = u[3 * k + 1] + (lane == 3 * k + 1 ? hstep : 0.0);
          const double w = u[3 * k + 2] + (lane == 3 * k + 2 ? hstep : 0.0);
          const double pdx = (vx * pcs[k] - vy * psn[k]) * p.dt, pdy = (vx * psn[k] + vy * pcs[k]) * p.dt;
          double prt = -2.0 * p.wo_n * (c.tyaw - th);
          if (k == n - 1) prt += -2.0 * p.wterm_o * (c.fyaw - th);
          SX += -2.0 * p.wt_n * (c.cx - x); SY += -2.0 * p.wt_n * (c.cy - y);
          ST += prt - pdy * SX + pdx * SY;
          // gradient entries of this lane's copy; the unperturbed lane 63 supplies the gradient itself
          // and the base of the forward difference
          const double g0 = p.dt * (pcs[k] * SX + psn[k] * SY), g1 = p.dt * (-psn[k] * SX + pcs[k] * SY),
                       g2 = p.dt * ST;
          if (kSteps) {  // (few variables: differencing after the sweep schedules better)
            hcol[3 * k] = g0; hcol[3 * k + 1] = g1; hcol[3 * k + 2] = g2;
          } else {       // (up to 24: difference at once, no second register array)
            const double b0 = lane_value(g0, 63), b1 = lane_value(g1, 63), b2 = lane_value(g2, 63);
            if (lane == 63) { gs[3 * k] = b0; gs[3 * k + 1] = b1; gs[3 * k + 2] = b2; }
            hc[3 * k] = (float)((g0 - b0) * inv_h);
            hc[3 * k + 1] = (float)((g1 - b1) * inv_h);
            hc[3 * k + 2] = (float)((g2 - b2) * inv_h);
          }
          x -= pdx; y -= pdy; th -= w * p.dt;
        } else {
          hc[3 * k] = 0.0f; hc[3 * k + 1] = 0.0f; hc[3 * k + 2] = 0.0f;
        }
      }
      if (kSteps) {
    #pragma unroll
        for (int j = 0; j < kVars; ++j) {
          const double base = lane_value(hcol[j], 63);
          if (lane == 63) gs[j] = base;
          hc[j] = (float)((hcol[j] - base) * inv_h);
        }
      }
      WAVE_SYNC();
    }
    else if (!kSteps || kRiccati) adjoint_by_scans<kTame, kRiccati, kFew>(a, c, L, exact_step, lane, n, free_path, nhops);
    else adjoint_short_sweep<kSteps, kTame>(a, c, L, have_trig, lane, n);
    NEO_PHASE(1);
    // ---- total gradient (control norm: minimal-norm subgradient at the kink), tangent-cone reduction at active bounds,
    //      face records of the second-order directions (tangent_cone.h)
    const bool my_corner = tangent_cone_pass<kTame, kNewton, kRiccati>(a, c, L, TOL[T_KINK], lane, n);
    WAVE_SYNC();
    // (in free space only -- no costmap term under the iterate's rollout: next to a cost step the Newton model is off either
    // way; the dense kernel knows that sum from the previous iteration's winner)
    const bool corner_any = kSecond && !kTame && __ballot(my_corner) != 0ull && (kRiccati ? free_path : (it > 0 && u_term == 0.0));
    NEO_PHASE(2);
    if (p.max_it == kDumpGradient) {
      // test hook (neo_mpc_gradient_batch): hand back the total gradient this kernel variant works with at
      // the projected x0 -- adjoint gradient of the smooth part + gradient of the control norm -- and stop
      for (int k = lane; k < nv; k += kLanes) a.solution[(size_t)b * nv + k] = gt[k];
      return true;
    }
    if (kSecond && it == 0 && cold) {
      // a cold start (x0 = 0, the reference's reset state py:359) is far from the minimiser and the
      // Newton step almost never wins there: steepest descent on the face for this one iteration
      if (kRiccati) { for (int k = lane; k < nv; k += kLanes) d[k] = -gr[k]; }
      else if (lane < nvr) d[lane] = -gr[lane];
      if (kRiccati) for (int i = lane; i < n; i += kLanes) AMODE[4 * i + 3] = 0;
      WAVE_SYNC();
    } else if (kRiccati) {
      riccati_prepare(a, c, L, n, lane, v_feasible, (float)TOL[T_MU]);
      WAVE_SYNC();
      if (corner_any) riccati_keep_linear_terms(a, L, n, lane, true);   // (the sweep writes its gains over them)
      auto sweep = [&]() {
        riccati_sweep<float, (kMinWavesPerSimd < 4), kFew>(a, L, n, lane);
        riccati_finish(a, c, L, n, lane);
      };
      sweep();
      if (!kTame && corner_any && repin_corner_blocks<true>(a, L, n, lane)) {   // (one-sided slides: once more, those blocks pinned)
        riccati_keep_linear_terms(a, L, n, lane, false);
        sweep();
      }
    } else if (kNewton) {
      newton_sol = dense_newton_direction<kSteps, kNwSteps, kTame>(a, L, hc, corner_any, lane, n, nvr);   // (dense_newton.h)
    }
    NEO_PHASE(3);
    // ---- projected L-BFGS: newest curvature pair, two-loop recursion on the reduced gradient, restriction to the face
    if (!kSecond) lbfgs_direction<kSteps, kPairs>(a, L, it, mem, head, npairs, lane, n);   // (lbfgs.h)
    if (kSecond && it > 0) {
      // the full Newton step is already below the step tolerance: u is the answer (blocks next to
      // the kink are moved by the prox step, which d does not describe -- keep iterating then)
      float dm = 0.0f;
      int anynear = 0;
      if (kRiccati) {
        // (a non-finite direction must not read as "no step left": fmaxf drops NaN)
        for (int k = lane; k < nv; k += kLanes) {
          const float v = (float)fabs(d[k]);
          dm = (v == v) ? fmaxf(dm, v) : INFINITY;
          anynear |= (AMODE[4 * (k / 3) + 2] == 1);
        }
      } else if (lane < nvr) { dm = fabsf(newton_sol); anynear = (AMODE[4 * (lane / 3) + 2] == 1); }   // (d[lane], still in a register)
      dm = wave_max_f(dm);
      const bool near_any = __ballot(anynear != 0) != 0ull;
      // (with a cheaper cell a hop away the search runs once more: its hop lanes decide)
      if ((double)dm < TOL[T_EARLY] && !near_any && nhops == 0) { status = NEO_MPC_STATUS_CONVERGED; break; }
      // a full Newton step below opt_tolerance (SLSQP's own step test) is the last one: searched
      // and taken like any other, but nothing re-checks the point it lands on
      // (a Gauss-Newton step converges linearly: it has to be shorter to be the last)
      if ((double)dm < ((kRiccati && !exact_step) ? kFinalFracGaussNewton : 1.0) * TOL[T_FINAL] && !near_any) final_step = true;
    }
    NEO_PHASE(4);
    if (p.max_it > kDumpGradient && it == p.max_it - kDumpGradient - 1) {
      // test hook (neo_mpc_direction_batch): the search direction of lanes 32-63 in this iteration
      for (int k = lane; k < nv; k += kLanes) a.solution[(size_t)b * nv + k] = d[k];
      return true;
    }
    // ---- 64 candidates, one rollout per lane; lowest objective wins
    // (run-time-sized Riccati kernel: the multiplier is re-read here -- one constant-memory load per iteration
    // against 2 x control_steps stages of work, and two registers fewer across the sweep)
    const double lscale = kRiccati ? lane_scale<kSecond>(lane) : my_scale;
    const double alpha_now = kFew ? TOL[T_ALPHA] : alpha;
    const double pstep = alpha_now * lscale;
    const double step = lane < 32 ? pstep : lscale;
    // Riccati, rollout in free space (no costmap term at any stage: the objective is smooth up to the control
    // norm's kink): the full Newton step -- lane 32's candidate -- is tried on its own first, one objective
    // evaluation with lane = stage, and taken without the 64-candidate search when it achieves kTrialRatio of
    // the decrease the quadratic model promises (-1/2 g_r . step).  Measured at control_steps 32 (8192 cold
    // starts, CPU mirror): 71 % of such trials succeed, 3.9 searches saved per solve for 0.3 iterations more;
    // the results are the same.  (With a costmap term under the rollout the search's spread of candidates is
    // what steps over cost edges and out of lethal cells.)
    constexpr double kTrialRatio = NEO_RULE_TRIAL_RATIO;
    bool took_trial = false;
    double fb = INFINITY, cterm = 0.0;   // (cterm, dense Newton: the costmap terms of this lane's candidate alone)
    int best = 32;
    if (kRiccati && free_path && it > 0) {
      const bool on = lane < n;
      double b0 = 0.0, b1 = 0.0, b2 = 0.0, g0 = 0.0, g1 = 0.0, g2 = 0.0;
      if (on) {
        candidate_block<kTame, kRiccati>(a, c, L, 32, 1.0, alpha_now, lane, b0, b1, b2);   // (lane 32: step length 1)
        g0 = gr[3 * lane]; g1 = gr[3 * lane + 1]; g2 = gr[3 * lane + 2];   // (u_new takes the reduced gradient's place)
        u_new[3 * lane] = b0; u_new[3 * lane + 1] = b1; u_new[3 * lane + 2] = b2;
      }
      auto scan = [](double v) { if constexpr (kFew > 0) return wave_scan_few<kFew>(v); else return wave_scan(v); };
      auto sum = [](double v) { if constexpr (kFew > 0) return wave_sum_few<kFew>(v); else return wave_sum(v); };
      const double th = scan(b2 * p.dt);
      double sn, cs;
      sincos_heading<kTame>(th, &sn, &cs);
      const double x = scan((b0 * cs - b1 * sn) * p.dt), y = scan((b0 * sn + b1 * cs) * p.dt);
      double fi = 0.0, pr = 0.0, tm = 0.0;   // (tm: the stage's costmap term alone)
      if (on) {
        const double dx = c.cx - x, dy = c.cy - y, et = c.tyaw - th;
        const double e0 = c.v0 - b0, e1 = c.v1 - b1, e2 = c.v2 - b2;
        tm = step_term<kCovered>(a, c, L, x, y);
        fi = p.wt_n * (dx * dx + dy * dy) + p.wo_n * (et * et) + p.wc_n * sqrt_fast(e0 * e0 + e1 * e1 + e2 * e2) + tm;
        if (lane == n - 1) { const double ef = c.fyaw - th; fi += p.wterm_o * (ef * ef); }
        pr = -0.5 * (g0 * (b0 - u[3 * lane]) + g1 * (b1 - u[3 * lane + 1]) + g2 * (b2 - u[3 * lane + 2]));
      }
      const double ft = sum(fi), pred = sum(pr);
      if (ft < f && f - ft >= kTrialRatio * pred) {
        took_trial = true; fb = ft;
        if (kBlockedRule) cterm = sum(tm);   // (the blocked-run rule asks whether the new iterate's rollout is free)
      }
      WAVE_SYNC();
    }
    // (first iteration only: the set-up armed the un-shifted previous solution as lane kAltLane's candidate)
    const double* ALT0 = L + a.lds.state + S_PREV_U0;
    bool alt_armed = false;
    if (it == 0) alt_armed = uniform_int(reinterpret_cast<const int*>(L + a.lds.state)[SI_HAS_PREV]) == kAltArmed;
    if (!took_trial) {
    // hop lanes (Riccati): lane h in 1..nhops tries the current point with the block of hop stage h - 1 changed
    int hop_stage = -1;
    float hop_x = 0.0f, hop_y = 0.0f;
    if (kRiccati && lane >= 1 && lane <= nhops) {
      const double* t = TOL;
      hop_stage = reinterpret_cast<const int*>(t + T_HOP_STAGE)[lane - 1];
      hop_x = reinterpret_cast<const float*>(t + T_HOP_VEC)[2 * (lane - 1)];
      hop_y = reinterpret_cast<const float*>(t + T_HOP_VEC)[2 * (lane - 1) + 1];
    }
    double fc = rollout_cost<kSteps, kTame, kCovered>(
        a, c, L,
        [&](int i, double& b0, double& b1, double& b2) {
          candidate_block<kTame, kRiccati>(a, c, L, lane, step, pstep, i, b0, b1, b2, hop_stage, hop_x, hop_y);
          // (a scalar branch taken in the first iteration only -- the empty asm keeps the compiler from turning it into
          // six selects per block that every iteration pays)
          if (it == 0) {
            asm volatile("");
            if (lane == 0) { b0 = u[3 * i]; b1 = u[3 * i + 1]; b2 = u[3 * i + 2]; }
            // (the un-shifted previous solution, armed by the set-up on the costmap: block 0 from the state slot)
            if (lane == kAltLane && alt_armed) { const double* src = i == 0 ? ALT0 : u + 3 * (i - 1); b0 = src[0]; b1 = src[1]; b2 = src[2]; }
          }
          if (kSteps) { cand[3 * i] = b0; cand[3 * i + 1] = b1; cand[3 * i + 2] = b2; }
        },
        [&](int i, double sn, double cs) {
          if (kSteps && !kSecond) { cand_sn[i] = sn; cand_cs[i] = cs; }
        }, kBlockedRule ? &cterm : nullptr);
    NEO_PHASE(5);
    if (!(fc == fc)) fc = INFINITY;
    if (it == 0) { f = lane_value(fc, 0); if (kNewton) u_term = lane_value(cterm, 0); }
    fb = fc;
    best = lane;
    wave_argmin(fb, best);
    }
    ++nfev;
    if (!(fb < f)) { status = NEO_MPC_STATUS_CONVERGED; ++it; break; }
    const bool alt_won = alt_armed && best == kAltLane;
    float stepmax = 0.0f;
    if (kSteps && kSecond && !took_trial) {
      // the winner holds its candidate in registers: it measures the step against u and overwrites u
      // in place -- one LDS round trip, no staging copy, no wave-wide maximum (the Newton paths keep
      // no previous iterate or gradient)
      if (lane == best) {
#pragma unroll
        for (int k = 0; k < 3 * kRegSteps; ++k) {
          stepmax = fmaxf(stepmax, (float)fabs(cand[k] - u[k]));
          u[k] = cand[k];
        }
      }
      stepmax = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, stepmax), best));
      have_trig = true;
    } else {
      if (kSteps && !kSecond) {
        if (lane == best) {
#pragma unroll
          for (int i = 0; i < kRegSteps; ++i) {
            u_new[3 * i] = cand[3 * i]; u_new[3 * i + 1] = cand[3 * i + 1]; u_new[3 * i + 2] = cand[3 * i + 2];
            ASN[i] = cand_sn[i]; ACS[i] = cand_cs[i];
          }
        }
      } else if (!kSteps && !took_trial) {
        // rebuild the winning candidate cooperatively: lane i takes control block i
        const double bstep = lane_value(step, best), bpstep = lane_value(pstep, best);
        int bhop = -1;
        float bhx = 0.0f, bhy = 0.0f;
        if (kRiccati && best >= 1 && best <= nhops) {   // (a hop candidate won)
          bhop = reinterpret_cast<const int*>(TOL + T_HOP_STAGE)[best - 1];
          bhx = reinterpret_cast<const float*>(TOL + T_HOP_VEC)[2 * (best - 1)];
          bhy = reinterpret_cast<const float*>(TOL + T_HOP_VEC)[2 * (best - 1) + 1];
        }
        for (int i = lane; i < n; i += kLanes) {
          double b0, b1, b2;
          candidate_block<kTame, kRiccati>(a, c, L, best, bstep, bpstep, i, b0, b1, b2, bhop, bhx, bhy);
          if (alt_won) { const double* src = i == 0 ? ALT0 : u + 3 * (i - 1); b0 = src[0]; b1 = src[1]; b2 = src[2]; }
          u_new[3 * i] = b0; u_new[3 * i + 1] = b1; u_new[3 * i + 2] = b2;
        }
      }
      have_trig = true;
      WAVE_SYNC();
      for (int k = lane; k < nv; k += kLanes) {
        const double nu = u_new[k], ou = u[k];
        stepmax = fmaxf(stepmax, (float)fabs(nu - ou));
        if (!kSecond) { u_prev[k] = ou; gt_prev[k] = gt[k]; }
        u[k] = nu;
      }
      stepmax = wave_max_f(stepmax);
    }
    const double gain = f - fb;
    // (gain thresholds are relative to the u-dependent part of the objective: inside the loop f excludes the constant
    // terms -- the terminal distance term, py:266, can be 20 x the rest)
    const double fscale = fmax(1.0, fabs(fb));
    stall = (gain <= TOL[T_FTOL] * fscale || (double)stepmax <= TOL[T_STALL]) ? stall + 1 : 0;
    // stage-wise direction: the window and closing-in rules only judge runs of BLOCKED iterations (none of the three won
    // by a Newton step of at least half its length); iterations won by the Newton step end through the step test
    // (the un-shifted start that won the first iteration says as little about step lengths as a hop)
    const bool hop_won = (kRiccati && best >= 1 && best <= nhops) || alt_won;
    nblocked = (best < 32 || lane_value(step, best) < kWindowStep || hop_won) ? nblocked + 1 : 0;
    const bool window_on = !kRiccati || kRouted || nblocked >= 3;
    // three iterations that together gained less than wtol: creeping along a costmap cell edge
    const double wtol0 = TOL[T_WTOL];   // (the closing-in rule below is on whenever the window rule is)
    const double wtol = it >= kLateIteration ? TOL[T_WTOL_LATE] : wtol0;
    const double g1 = kFew ? TOL[T_GAIN1] : gain1, g2 = kFew ? TOL[T_GAIN2] : gain2;   // (the previous two iterations' gains)
    bool creeping = wtol > 0.0 && gain + g1 + g2 <= wtol * fscale && window_on;
    // ... and so does a step below stall_step whose gain halved twice in a row: the search is closing in on a
    // costmap cell edge (or the kink) geometrically; what is left to gain is less than the last gain (in free space three
    // gains it takes: round 5 -- the INFINITY the two older ones start at used to pass for a gain, and a warm search that began next to the
    // kink ended after its second iteration, 2.4e-3 from the reference's converged first control on one G13 tick -- and the
    // geometric series the gains start has to be worth less than the stall threshold, gain r / (1 - r) <= ftol f~ with r =
    // gain / gain1: 3e-6 left to gain is 2.5e-3 in the first control along a direction of curvature 1)
    // (dense and L-BFGS directions: only behind kClosingRun blocked iterations -- the gains of a Newton search that converges
    // quadratically halve twice in a row as well, and one blocked iteration after them, a bound about to become active, is
    // no sign of creeping: random parameter sets, 1 solve in 3000 stopped 5e-3 short)
    // (in free space -- where the first control is gated, not only the objective: dense direction: no costmap term under the
    // NEW iterate's rollout; stage-wise: under the rollout the iteration started from)
    bool free_now = kRiccati ? free_path : false;
    if (kNewton) free_now = lane_value(cterm, best) == 0.0;
    creeping = creeping || (wtol0 > 0.0 && (double)stepmax <= TOL[T_STALL] && gain <= 0.5 * g1 && g1 <= 0.5 * g2 &&
                            (!free_now || (g2 < INFINITY && gain * gain <= TOL[T_FTOL] * fscale * (g1 - gain))) &&
                            ((kRiccati && !kRouted) ? window_on : nblocked >= kClosingRun));
    // Blocked-run stop rule (dense Newton).  kBlockedRun iterations in a row not won by a decent Newton step that
    // together gain less than 0.1 x opt_tolerance (0.03 x with no costmap term under the new iterate's rollout): something
    // the quadratic model does not see is in the way -- a costmap cell edge, or blocks hovering next to the control norm's
    // kink -- and the search advances 1e-6 of f per iteration (SLSQP stops on ONE iteration gaining less than
    // opt_tolerance).  In a closed 30 Hz loop of 4096 robots such searches set the duration of every launch: per-tick
    // maximum 25 -> 13 iterations in the median, 100 -> 16 at worst; cold solves and the zero-map drift check are untouched.
    bool blocked_stop = false;
    if (kBlockedRule) {
      const double bs = lane_value(step, best);
      blocked_run = (best < 32 || bs < kBlockedStep) ? blocked_run + 1 : 0;
      if (blocked_run >= kBlockedRun) {
        const bool free_rollout = lane_value(cterm, best) == 0.0;
        const double btol = free_rollout ? TOL[T_BTOL_FREE] : TOL[T_BTOL_MAP];
        blocked_stop = gain + g1 + g2 <= btol;   // (btol 0: the rule is off -- a gain is never <= 0 here)
      }
    }
    double* TOLW = L + tol_off;
    if (kFew) { if (lane == 0) { TOLW[T_GAIN2] = g1; TOLW[T_GAIN1] = gain; } }
    else { gain2 = gain1; gain1 = gain; }
    f = fb;
    // the last-step rule rests on the Newton model having held: an iteration announced as the last but WON by a proximal
    // step or a short Newton step (a bound about to become active, the kink) is not the last
    if (nblocked != 0) final_step = false;
    if (kNewton) u_term = lane_value(cterm, best);
    // (a hop that won says nothing about step lengths: damping and proximal step stay as they are)
    if (kRiccati && !(it == 0 && cold) && !hop_won) {   // (an iteration that had a Newton direction)
      // (step lengths of the Newton lanes as lane masks, feasible_set.h: scalar bit tests)
      constexpr unsigned long long kNearlyFull = newton_lanes_at_least(0.8), kShort = kNewtonLanes & ~newton_lanes_at_least(0.3);
      const double mu0 = n > 8 ? (double)(n - 8) * 0.125 : 0.0, mu = TOL[T_MU];
      double* mu_slot = L + tol_off + T_MU;
      if ((kNearlyFull >> best) & 1ull) { if (lane == 0) *mu_slot = fmax(0.25 * mu, mu0 * 0.0625); }
      else if (best < 32 || ((kShort >> best) & 1ull)) { if (lane == 0) *mu_slot = fmin(4.0 * mu, 16.0 * mu0); }
    }
    if (best < 32 && !hop_won) {
      const double al = clampd(lane_value(step, best), 1e-6, 1e6);
      if (kFew) { if (lane == 0) TOLW[T_ALPHA] = al; }
      else alpha = al;
    }
    WAVE_SYNC();
    NEO_PHASE(6);
    NEO_PHASE_DUMP();
    if ((double)stepmax < TOL[T_XTOL] || stall >= kStallIterations || creeping || final_step || blocked_stop) { status = NEO_MPC_STATUS_CONVERGED; ++it; break; }
  }

  // ---- phase 2, second-order directions: a search that has ENDED looks once at the costmap cells around every stage
  //      (cell_scan.h): cheaper cells up to three cells away, which no descent direction sees -- the term has no gradient --
  //      and SLSQP's line search samples by accident.  Skipped when the whole reach tile is free or (dense direction: known
  //      from the winner's rollout) no stage of the iterate has a costmap term under it.  Behind a scan that gained more
  //      than opt_tolerance the search is taken up again: the other blocks have a new neighbour to adjust to.
  // (round 6) A scan that found a cheaper cell is followed by another from where it has put the iterate (one pass of this loop
  // each, NEO_RULE_SCAN_REPEATS at most), until a scan finds nothing: the returned point is a fixed point of the scan -- solved
  // again from its own answer an instance used to get a second look and move.  The round as a whole decides about the resume.
  if (!kSecond || scanned || status != NEO_MPC_STATUS_CONVERGED || p.max_it >= kDumpGradient) break;
  const bool nothing_to_scan = (c.tile_geom & kTileFree) || (kNewton && u_term == 0.0);
  if (nothing_to_scan && nscans == 0) break;
  bool resume = false;
  {
    SolveArgs as;
    fresh_args<kSteps, kStaticTile, kLayoutSteps, kRouted>(as);
    select_map(as.map, L + as.lds.prob);
    Ctx cs;
    ctx_from_lds<false>(as, L, cs);
    int ls = lane_again();
    if (nscans == 0 && ls == 0) L[as.lds.tol + T_FSCAN] = f;
    NEO_SEGMENT_SCAN_BEGIN();
    const bool w = !nothing_to_scan &&
                   cell_scan<kSteps, kTame, kCovered>(as, cs, L, f, kNewton ? &u_term : nullptr, nfev, ls, kSteps ? kSteps : as.p.n);
    NEO_SEGMENT_SCAN_END();
    ++nscans;
    scan_only = w && nscans < NEO_RULE_SCAN_REPEATS;
    if (!scan_only) {
      scanned = true;
      WAVE_SYNC();
      resume = L[as.lds.tol + T_FSCAN] - f > as.p.scan_resume_gain && it < as.p.max_it;
      if (kFew && resume) {
        if (ls == 0) { L[as.lds.tol + T_GAIN1] = INFINITY; L[as.lds.tol + T_GAIN2] = INFINITY; }
        WAVE_SYNC();
      }
    }
  }
  if (scan_only) continue;
  if (!resume) break;
  status = NEO_MPC_STATUS_MAX_ITER; stall = 0; final_step = false; blocked_run = 0; nblocked = 1;
  gain1 = INFINITY; gain2 = INFINITY;
  }
  NEO_SEGMENT(1);
  NEO_SEGMENT_DUMP();
  // (a search taken up again behind a scan that runs into the iteration cap HAD converged, and the scan only improved its
  // point: it is reported converged -- status 1 would hand the warm start back un-shifted, py:399-400, for a better answer)
  if (scanned && status == NEO_MPC_STATUS_MAX_ITER) status = NEO_MPC_STATUS_CONVERGED;
  sc.f = f; sc.it = it; sc.nfev = nfev; sc.status = status;
  return false;
}

// ---- phase 3: K2 (py:365-403) on the search's result
template <int kSteps, int kStaticTile, int kLayoutSteps>
__device__ __forceinline__ void solve_finish(double* L, uint32_t b, int lane, const SolveCarry& sc) {
  const int flags = sc.flags, status = sc.status, it = sc.it, nfev = sc.nfev;
  double f = sc.f;
  SolveArgs a;
  fresh_args<kSteps, kStaticTile, kLayoutSteps>(a);
  select_map(a.map, L + a.lds.prob);
  const DevParams& p = a.p;
  const int n = kSteps ? kSteps : p.n, nv = 3 * n;
  double* u = L + a.lds.u;
  Ctx c;
  ctx_from_lds<false>(a, L, c);
#ifndef NEO_MPC_PHASE_TIMING
  // (test hooks: a search that ended before the dumped iteration keeps the NaN row the host put there)
  if (a.solution && p.max_it < kDumpGradient)
    for (int k = lane; k < nv; k += kLanes) a.solution[(size_t)b * nv + k] = u[k];
#endif
  WAVE_SYNC();
  f += L[a.lds.tol + T_KONST];
  c.true_yaw = L[a.lds.tol + T_TRUE_YAW];
  postprocess(a, c, L, b, lane, u, status == NEO_MPC_STATUS_CONVERGED, L[a.lds.prob + P_FOOTPRINT], flags, f, status, it, nfev);
}

// ---------------------------------------------------------------- K1: the kernels
// k_solve<waves per SIMD, control_steps specialisation, direction, tame, static tile>: one direction for every instance.
template <int kMinWavesPerSimd, int kSteps, int kDir = 0, bool kTame = false, int kStaticTile = 0>
__global__ __launch_bounds__(kLanes, kMinWavesPerSimd) void k_solve(const SolveArgs args) {
  extern __shared__ __align__(16) double Ldyn[];
  // (the run-time-sized Newton kernel carves LDS for its largest system and no L-BFGS pairs)
  constexpr int kLayoutSteps = kSteps ? kSteps : kNewtonMaxSteps;
  constexpr LdsLayout kStaticLayout = make_lds_layout(kLayoutSteps, kSteps ? 4 : 0, false);
  constexpr int kStaticDoubles = kStaticTile ? (kStaticLayout.total_bytes + kStaticTile) / 8 : 2;
  __shared__ __align__(16) double Lstat[kStaticDoubles];
  double* const L = kStaticTile ? Lstat : Ldyn;
  const int lane = threadIdx.x;
  if (blockIdx.x >= args.count) return;
  // (balanced dispatch, neo_mpc_balance_dispatch_device: which instance this workgroup solves -- instances are independent,
  // every result is bit for bit what it is in launch order; only which of them share a SIMD changes)
  const uint32_t b = args.order ? (uint32_t)__builtin_amdgcn_readfirstlane((int)args.order[blockIdx.x]) : blockIdx.x;
  if (b >= args.count) return;   // (an order the caller launched this solve ahead of, on another stream: never out of range)
  NEO_WAVE_START;
  SolveCarry sc;
  if (!solve_setup<kSteps, kTame, kStaticTile, kLayoutSteps>(L, b, lane, sc)) return;
  if (solve_search<kMinWavesPerSimd, kSteps, kDir, kTame, kStaticTile, kLayoutSteps>(L, b, sc)) return;
  solve_finish<kSteps, kStaticTile, kLayoutSteps>(L, b, lane, sc);
  NEO_WAVE_END_ARGS(args);
}

#ifdef NEO_MPC_TU_RICCATI   // (built without the SLP vectoriser like the stage-wise variants: with it the stage-wise branch spills 22 vector registers)
// k_solve_routed: control_steps 3, AUTO below the heavy-costmap threshold -- DIRECTION BY NEIGHBOURHOOD (round 6;
// solver_rules.h neo_rules_routes_by_neighbourhood).  The set-up stages the reach tile and knows whether a LETHAL cell is among
// its cells.  No wall in reach (88 % of the BASELINE config-2 instances): the dense projected Newton search (registers,
// finite-difference Hessian) with the cell scan behind it -- round 5's headline kernel.  A wall in reach: the stage-wise
// (Riccati) search, whose wall model slides along walls -- every objective miss the random-parameter fuzz against the reference
// found at control_steps 3 was a dense search hemmed in by lethal cells.  One launch, one wave per instance, one wave-uniform
// branch behind the set-up; the two branches share LDS (the stage-wise carve-up for three stages lies inside the dense one,
// fresh_args) and the register budget (128 at four waves per SIMD, no scratch).
template <int kMinWavesPerSimd, bool kTame = false, int kStaticTile = 0>
__global__ __launch_bounds__(kLanes, kMinWavesPerSimd) void k_solve_routed(const SolveArgs args) {
  extern __shared__ __align__(16) double Ldyn[];
  constexpr int kSteps = kRoutedSteps;
  constexpr LdsLayout kStaticLayout = make_lds_layout(kSteps, 4, false);
  constexpr int kStaticDoubles = kStaticTile ? (kStaticLayout.total_bytes + kStaticTile) / 8 : 2;
  __shared__ __align__(16) double Lstat[kStaticDoubles];
  double* const L = kStaticTile ? Lstat : Ldyn;
  const int lane = threadIdx.x;
  if (blockIdx.x >= args.count) return;
  const uint32_t b = args.order ? (uint32_t)__builtin_amdgcn_readfirstlane((int)args.order[blockIdx.x]) : blockIdx.x;
  if (b >= args.count) return;
  NEO_WAVE_START;
  SolveCarry sc;
  if (!solve_setup<kSteps, kTame, kStaticTile, kSteps>(L, b, lane, sc)) return;
  // (the test hooks dump what the instance's own direction works with)
  if (__builtin_amdgcn_readfirstlane((int)sc.wall_in_reach)) {
    // (the stage-wise searches are the long ones -- 7.3 iterations against 5.1, up to 19 against 11, 1.5 x the instructions per
    // iteration -- and a launch of 4096 ends with the longest of them: they issue ahead of the dense waves they share a SIMD
    // with, which finish early either way.  Same box, three runs each: 23.0 -> 24.5 M solves/s; 262 144 instances: no change)
    __builtin_amdgcn_s_setprio(3);
    if (solve_search<kMinWavesPerSimd, kSteps, 2, kTame, kStaticTile, kSteps, true>(L, b, sc)) return;
  } else {
    if (solve_search<kMinWavesPerSimd, kSteps, 1, kTame, kStaticTile, kSteps, false>(L, b, sc)) return;
  }
  solve_finish<kSteps, kStaticTile, kSteps>(L, b, lane, sc);
  NEO_WAVE_END_ARGS(args);
}
#endif

#ifndef NEO_MPC_TU_RICCATI   // (neo_mpc_riccati.hip compiles this file again for the Riccati variants of K1 only)
// K2 on its own: `solution` supplies x.x, `success` supplies x.success
__global__ __launch_bounds__(kLanes) void k_postprocess(const SolveArgs args) {
  extern __shared__ __align__(16) double L[];
  SolveArgs a = args;
  const int lane = threadIdx.x;
  const uint32_t b = blockIdx.x;
  if (b >= a.count) return;
  const int nv = 3 * a.p.n;
  load_records(a, L, b, lane);
  if (uniform_int(reinterpret_cast<const int*>(L + a.lds.prob)[PI_SKIP]) == 1) { skip_instance(a, b, lane, true); return; }   // (no request this tick: see k_solve)
  select_map(a.map, L + a.lds.prob);
  int flags = reset_and_warm(a, L, b, lane) ? NEO_MPC_FLAG_RESET : 0;
  const double fcost = footprint_cost(a, L, b, lane);
  Ctx c;
  make_ctx_wave(a.p, a.map, L + a.lds.prob, fcost, c, lane);
  load_tile(a, c, L, lane);
  if (c.tile_geom & kTileWall) flags |= NEO_MPC_FLAG_WALL_IN_REACH;
  double* u = L + a.lds.u;
  for (int k = lane; k < nv; k += kLanes) u[k] = a.solution[(size_t)b * nv + k];
  WAVE_SYNC();
  double f = rollout_cost(a, c, L, [&](int i, double& b0, double& b1, double& b2) {
    b0 = u[3 * i]; b1 = u[3 * i + 1]; b2 = u[3 * i + 2];
  });
  f = __shfl(f, 0);
  const bool success = a.success ? a.success[b] != 0 : true;
  postprocess(a, c, L, b, lane, u, success, fcost, flags, f, success ? 0 : 1, 0, 1);
}

// py:204-269 for given controls, one lane per instance -- the PARITY kernel: it follows the reference statement by
// statement (the solver's own rollout forms the world position as X0 + Rot(psi0) (x, y) from the base-frame rollout;
// this one accumulates odom_yaw, pos_x and pos_y step by step like py:234-236, takes the square root of the distance and
// squares it again like py:250-252, and divides by control_steps where the reference does), so that it differs from the
// reference only in the last bits of sin / cos / atan2.
__global__ __launch_bounds__(256) void k_objective(const ObjectiveArgs a) {
  __shared__ double cost_of[256];   // getCost by raw cell value: occupancy / 100 (the build's costmap contract)
  cost_of[threadIdx.x] = raw_cost((int)threadIdx.x);
  __syncthreads();
  const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= a.count) return;
  DevMap map = a.map;
  const double* P = reinterpret_cast<const double*>(a.problems + b);
  select_map<false>(map, P);
  const int n = a.p.n;
  const double dt = a.p.dt, w_trans = a.w_trans, w_orient = a.w_orient, w_control = a.w_control;
  const double target_yaw = yaw_of(P + P_CARROT_Q);                       // py:211
  const double final_yaw = yaw_of(P + P_GOAL_Q);                          // py:212
  const double q0[4] = {P[P_CUR_Q], P[P_CUR_Q + 1], P[P_CUR_Q + 2],
                        (a.p.compat & NEO_MPC_COMPAT_ODOM_YAW_GOAL_W) ? P[P_GOAL_Q + 3] : P[P_CUR_Q + 3]};
  double odom_yaw = yaw_of(q0);                                           // py:213 (the goal's w: reference quirk)
  const double* u = a.u + (size_t)b * 3 * n;
  const double footprint = P[P_FOOTPRINT];
  double total = 0.0, x = 0.0, y = 0.0, z = 0.0;
  double pos_x = P[P_CUR_X], pos_y = P[P_CUR_Y];                          // py:220-221
  for (int i = 0; i < n; ++i) {
    const double vx = u[3 * i], vy = u[3 * i + 1], wz = u[3 * i + 2];
    z += wz * dt;                                                         // py:230
    double sz, cz;
    sincos(z, &sz, &cz);
    x += (vx * cz * dt - vy * sz * dt);                                   // py:231
    y += (vx * sz * dt + vy * cz * dt);                                   // py:232
    odom_yaw += wz * dt;                                                  // py:234
    double so, co;
    sincos(odom_yaw, &so, &co);
    pos_x += vx * co * dt - vy * so * dt;                                 // py:235
    pos_y += vx * so * dt + vy * co * dt;                                 // py:236
    const int mx = cell_of(pos_x, map.origin_x, map.resolution, map.inv_resolution);   // py:246
    const int my = cell_of(pos_y, map.origin_y, map.resolution, map.inv_resolution);
    const double c = cost_of[map_raw(map, mx, my)];
    const double costmap_cost = c * c;                                    // py:247
    const double ddx = P[P_CARROT_X] - x, ddy = P[P_CARROT_Y] - y;
    const double dist = sqrt(ddx * ddx + ddy * ddy);                      // py:250
    const double eth = target_yaw - z;                                    // py:251
    total += ((w_trans * (dist * dist)) + (w_orient * (eth * eth))) / n;  // py:252
    const double e0 = P[P_VEL] - vx, e1 = P[P_VEL + 1] - vy, e2 = P[P_VEL + 2] - wz;
    total += w_control * sqrt(e0 * e0 + e1 * e1 + e2 * e2) / n;           // py:253-254
    if (c == 1.0) total += costmap_cost * 1000 / n;                       // py:257-258
    else total += a.w_costmap * costmap_cost / n;                         // py:260
    if (footprint == 1.0) total += (footprint * footprint) * a.p.w_footprint / n;   // py:262-263
  }
  const double gdx = P[P_CARROT_X] - P[P_GOAL], gdy = P[P_CARROT_Y] - P[P_GOAL + 1];
  const double gdist = sqrt(gdx * gdx + gdy * gdy);                       // py:266
  const double eth = final_yaw - z;                                       // py:267
  total += ((w_trans * (gdist * gdist)) + (w_orient * (eth * eth))) * a.w_terminal;   // py:268
  a.cost[b] = total;
}

// K3: raw nav2 costmap -> bordered, pitched device map.  One 16-byte store per lane and chunk; kIngestUnroll chunks per
// thread with every load issued before the first store (memory-level parallelism: the kernel is a pure stream).
constexpr int kIngestUnroll = 4;
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ u32x4 ingest_chunk(const IngestArgs& a, unsigned idx, unsigned chunks_per_row) {
  const int row = (int)(idx / chunks_per_row), chunk = (int)(idx - (unsigned)row * chunks_per_row);
  const int my = row - a.border;
  const int mx0 = chunk * 16 - a.border;
  u32x4 v = {0xFEFEFEFEu, 0xFEFEFEFEu, 0xFEFEFEFEu, 0xFEFEFEFEu};
  if (my >= 0 && my < a.size_y && mx0 >= 0 && mx0 + 16 <= a.size_x && (a.size_x & 7) == 0) {
    // interior chunk of a map whose rows are 8-byte aligned (mx0 is a multiple of 16): two 8-byte loads
    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
    const u32x2* src8 = reinterpret_cast<const u32x2*>(a.src + (long)my * a.size_x + mx0);
    const u32x2 lo = __builtin_nontemporal_load(src8), hi = __builtin_nontemporal_load(src8 + 1);
    v = u32x4{lo.x, lo.y, hi.x, hi.y};
  } else if (my >= 0 && my < a.size_y && mx0 >= 0 && mx0 + 16 <= a.size_x && (a.size_x & 3) == 0) {
    const uint32_t* src4 = reinterpret_cast<const uint32_t*>(a.src + (long)my * a.size_x + mx0);
    v = u32x4{src4[0], src4[1], src4[2], src4[3]};
  } else if (my >= 0 && my < a.size_y && mx0 >= 0 && mx0 < a.size_x && (a.size_x & 3) == 0) {
    // the chunk that straddles the right edge of a map whose width is a multiple of 4 (200-cell windows: 8 of its 16
    // bytes): whole dwords, lethal beyond the edge
    const uint32_t* src4 = reinterpret_cast<const uint32_t*>(a.src + (long)my * a.size_x + mx0);
    const int valid = (a.size_x - mx0) >> 2;   // 1..3 dwords
    v = u32x4{src4[0], valid > 1 ? src4[1] : 0xFEFEFEFEu, valid > 2 ? src4[2] : 0xFEFEFEFEu, 0xFEFEFEFEu};
  } else if (my >= 0 && my < a.size_y && mx0 + 16 > 0 && mx0 < a.size_x) {
    uint8_t bytes[16];
    const uint8_t* src = a.src + (long)my * a.size_x;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const int mx = mx0 + k;
      bytes[k] = (mx >= 0 && mx < a.size_x) ? src[mx] : (uint8_t)254;
    }
    v = *reinterpret_cast<const u32x4*>(bytes);
  }
  return v;
}
__global__ __launch_bounds__(256) void k_ingest(const IngestArgs args) {
  IngestArgs a = args;   // blockIdx.y: which map of a pool
  a.src += (long)blockIdx.y * a.size_x * a.size_y;
  a.dst += (long)blockIdx.y * a.dst_stride;
  const unsigned chunks_per_row = (unsigned)a.pitch >> 4;
  const unsigned total = (unsigned)a.rows * chunks_per_row;   // (< 2^31: rows, pitch <= 2^20 + 256 and pitch/16 per row)
  const unsigned stride = gridDim.x * blockDim.x;
  for (unsigned base = blockIdx.x * blockDim.x + threadIdx.x; base < total; base += kIngestUnroll * stride) {
    u32x4 v[kIngestUnroll];
#pragma unroll
    for (int k = 0; k < kIngestUnroll; ++k)
      if (base + k * stride < total) v[k] = ingest_chunk(a, base + k * stride, chunks_per_row);
    // streamed once, read back sparsely (reach tiles): non-temporal, so the stream does not wait for
    // L2 lines to be allocated (measured: 3.0 -> 4.3 TB/s over a pool of 4096 windows)
#pragma unroll
    for (int k = 0; k < kIngestUnroll; ++k)
      if (base + k * stride < total)
        __builtin_nontemporal_store(v[k], reinterpret_cast<u32x4*>(a.dst) + (base + k * stride));
  }
}

// K4: carrot selection, one wavefront per robot.  Lanes stride over the plan poses (24 B each,
// consecutive lanes read consecutive poses: one contiguous 1.5 KB segment per wave load), so the
// kernel streams the plans once for the closest-pose search (cpp:83-88) and re-reads only the
// kept window [begin, end) for the cut-off and look-ahead scans (cpp:102-106, 177-186).
__global__ __launch_bounds__(kLanes) void k_carrot(const CarrotArgs a) {
  const int lane = threadIdx.x;
  const size_t b = blockIdx.x;
  if (b >= a.b.count) return;
  const uint32_t o0 = a.b.plan_offsets[b], np = a.b.plan_offsets[b + 1] - o0;
  const double* poses = a.b.plan_poses + 3 * (size_t)o0;
  const double rx = a.b.robot_poses[3 * b], ry = a.b.robot_poses[3 * b + 1], rth = a.b.robot_poses[3 * b + 2];
  int slow = a.b.slow_down[b];
  neo_mpc_carrot out;
  out.xy[0] = 0.0; out.xy[1] = 0.0; out.q[0] = 0.0; out.q[1] = 0.0; out.q[2] = 0.0; out.q[3] = 1.0;
  out.lookahead_dist = 0.0; out.begin = 0; out.end = 0; out.closer_to_goal = 0; out.slow_down = slow;
  out.status = 0; out.reserved = 0;
  if (np == 0) {                                                        // cpp:69-71
    out.status = 1;
    if (lane == 0) { a.b.carrots[b] = out; if (a.b.problems) a.b.problems[b].skip = 1; }   // (a throw: no request this tick)
    return;
  }
  // closest pose, first minimum (min_by, cpp:83-88)
  double best = INFINITY;
  uint32_t besti = 0xffffffffu;
  uint32_t k = lane;
  for (; k + 3 * kLanes < np; k += 4 * kLanes) {  // four independent pose loads in flight per lane
#define NT(p) __builtin_nontemporal_load(p)
    const double x0 = NT(poses + 3 * k), y0 = NT(poses + 3 * k + 1);
    const double x1 = NT(poses + 3 * (k + kLanes)), y1 = NT(poses + 3 * (k + kLanes) + 1);
    const double x2 = NT(poses + 3 * (k + 2 * kLanes)), y2 = NT(poses + 3 * (k + 2 * kLanes) + 1);
    const double x3 = NT(poses + 3 * (k + 3 * kLanes)), y3 = NT(poses + 3 * (k + 3 * kLanes) + 1);
#undef NT
    const double d0 = hypot(x0 - rx, y0 - ry), d1 = hypot(x1 - rx, y1 - ry);
    const double d2 = hypot(x2 - rx, y2 - ry), d3 = hypot(x3 - rx, y3 - ry);
    if (d0 < best) { best = d0; besti = k; }
    if (d1 < best) { best = d1; besti = k + kLanes; }
    if (d2 < best) { best = d2; besti = k + 2 * kLanes; }
    if (d3 < best) { best = d3; besti = k + 3 * kLanes; }
  }
  for (; k < np; k += kLanes) {
    const double d = hypot(poses[3 * k] - rx, poses[3 * k + 1] - ry);
    if (d < best) { best = d; besti = k; }
  }
  const double gmin = wave_min(best);
  const uint32_t begin = (uint32_t)wave_min((best == gmin) ? (double)besti : 4.0e9);
  out.closer_to_goal = hypot(poses[3 * (np - 1)] - rx, poses[3 * (np - 1) + 1] - ry) <=
                       a.lp.lookahead_dist_close_to_goal ? 1 : 0;       // cpp:95-100
  double la = a.lp.lookahead_dist_min;                                  // cpp:161-169
  if (!slow || out.closer_to_goal) {
    la = a.lp.lookahead_dist_max;
    if (out.closer_to_goal) la = a.lp.lookahead_dist_close_to_goal;
  }
  double sn, cs;
  sincos(rth, &sn, &cs);
  // one scan from `begin`: first pose beyond the costmap (cpp:102-106) and first pose at least the
  // look-ahead distance away in the base frame (cpp:177-181)
  uint32_t end = np, pick = 0xffffffffu;
  for (uint32_t base = begin; base < np; base += kLanes) {
    const uint32_t k = base + lane;
    bool far = false, hit = false;
    if (k < np) {
      const double dx = poses[3 * k] - rx, dy = poses[3 * k + 1] - ry;
      far = hypot(dx, dy) > a.lp.max_transform_dist;
      hit = hypot(cs * dx + sn * dy, -sn * dx + cs * dy) >= la;
    }
    const unsigned long long mfar = __ballot(far), mhit = __ballot(hit);
    if (pick == 0xffffffffu && mhit) pick = base + (uint32_t)__ffsll((long long)mhit) - 1;
    if (mfar) { end = base + (uint32_t)__ffsll((long long)mfar) - 1; break; }
  }
  out.begin = begin; out.end = end;
  if (end == begin) {                                                   // cpp:130-132
    out.status = 2;
    if (lane == 0) { a.b.carrots[b] = out; if (a.b.problems) a.b.problems[b].skip = 1; }
    return;
  }
  out.lookahead_dist = la;
  if (pick == 0xffffffffu || pick >= end) pick = end - 1;               // cpp:183-186
  const double dx = poses[3 * pick] - rx, dy = poses[3 * pick + 1] - ry;
  const double yaw_local = poses[3 * pick + 2] - rth;
  out.xy[0] = cs * dx + sn * dy; out.xy[1] = -sn * dx + cs * dy;
  double qs, qc;
  sincos(0.5 * yaw_local, &qs, &qc);
  out.q[2] = qs; out.q[3] = qc;
  const double cy = fabs(atan2(2.0 * qc * qs, 1.0 - 2.0 * qs * qs));  // createYawFromQuat (cpp:54-62)
  slow = (cy >= 1.0 && a.b.footprint_costs && a.b.footprint_costs[b] > 200.0) ? 1 : 0;   // cpp:221-232
  out.slow_down = slow;
  // cpp:234-236: a footprint cost of 255 makes the plugin throw here -- after the slow_down_ update, before the
  // optimizer request: this robot makes no request this tick
  const bool no_request = a.b.footprint_costs && a.b.footprint_costs[b] == 255.0;
  if (no_request) out.status = 3;
  if (lane == 0) {
    a.b.carrots[b] = out;
    a.b.slow_down[b] = slow;
    if (a.b.problems) {
      neo_mpc_problem* pr = a.b.problems + b;
      pr->carrot_xy[0] = out.xy[0]; pr->carrot_xy[1] = out.xy[1];
      pr->carrot_q[0] = 0.0; pr->carrot_q[1] = 0.0; pr->carrot_q[2] = qs; pr->carrot_q[3] = qc;
      pr->switch_opt = out.closer_to_goal;   // cpp:245
      pr->skip = no_request ? 1 : 0;
    }
  }
}
#endif  // NEO_MPC_TU_RICCATI

}  // namespace

// Register budget of K1: with the per-iteration opaque lane index (see the top of the solver loop)
// the Newton kernel needs 124 VGPRs and the generic kernel 118, both spill-free at 4 waves/SIMD; the
// control_steps == 3 L-BFGS kernel needs 130 (3 waves/SIMD).  __launch_bounds__ pins the occupancy
// each was measured at, so that a later edit cannot silently drop a wave per SIMD (it would spill
// instead, which `make resource-usage` shows).  NEO_MPC_SOLVE_WAVES=2|3|4 in the environment of neo_mpc_create overrides
// (LaunchTuning), for A/B runs.
static int solve_variant(const LaunchTuning& t, int fallback) { return t.solve_waves ? t.solve_waves : fallback; }

// The Riccati variants of K1 live in a translation unit of their own (neo_mpc_riccati.hip = this file with
// NEO_MPC_TU_RICCATI, compiled with -fno-slp-vectorize): the SLP vectoriser packs the sweep's float32
// arithmetic into v_pk_* instructions and pays for it with three hundred register moves that assemble the
// operand pairs (859 vector instructions in the sweep against 757 without it) -- measured +7 % solves/s at
// control_steps 8 and +9 % at 32 without; the dense-Newton kernels are 0.5 % faster WITH it.
void launch_solve_riccati(const SolveArgs& a, const LaunchTuning& tuning, void* stream, void* ev_start, void* ev_stop);

#ifdef NEO_MPC_TU_RICCATI
void launch_solve_riccati(const SolveArgs& a, const LaunchTuning& tuning, void* stream, void* ev_start, void* ev_stop) {
#else
void launch_solve(const SolveArgs& a, const LaunchTuning& tuning, void* stream, void* ev_start, void* ev_stop) {
#endif
  if (a.count == 0) return;
  const dim3 grid(a.count), block(kLanes);
  hipStream_t st = (hipStream_t)stream;
  const bool generic = a.p.mem != 4;  // (the control_steps-3 L-BFGS specialisation carves LDS for four pairs)
  const bool disc = a.p.tame != 0 && !tuning.no_tame;
  const size_t lds = a.lds.total_bytes;
  // (the static-tile kernels count on the tile being there: no tile at all -- a reach of 60 cells and more -- is not "small")
  const bool small_tile = a.lds.tile_w * a.lds.tile_h > 0 && a.lds.tile_w * a.lds.tile_h <= 1024 && !tuning.dynamic_lds;
  // with events: hipExtLaunchKernel stamps them from the dispatch packet itself (no barrier packets in
  // front of and behind the kernel, which is what separate hipEventRecord calls put on the queue)
  hipEvent_t e0 = (hipEvent_t)ev_start, e1 = (hipEvent_t)ev_stop;
#define NEO_LAUNCH_LDS(bytes, ...)                                                                       \
  do {                                                                                                   \
    if (e0 || e1) hipExtLaunchKernelGGL((k_solve<__VA_ARGS__>), grid, block, (bytes), st, e0, e1, 0, a);   \
    else hipLaunchKernelGGL((k_solve<__VA_ARGS__>), grid, block, (bytes), st, a);                          \
  } while (0)
#define NEO_LAUNCH(...) NEO_LAUNCH_LDS(lds, __VA_ARGS__)
#define NEO_LAUNCH_ROUTED(bytes, ...)                                                                    \
  do {                                                                                                   \
    if (e0 || e1) hipExtLaunchKernelGGL((k_solve_routed<__VA_ARGS__>), grid, block, (bytes), st, e0, e1, 0, a);   \
    else hipLaunchKernelGGL((k_solve_routed<__VA_ARGS__>), grid, block, (bytes), st, a);                          \
  } while (0)
#define NEO_LAUNCH_ROUTED_W(w, tame)                                                                     \
  do {                                                                                                   \
    if ((w) == 4) NEO_LAUNCH_ROUTED(lds, 4, tame); else if ((w) == 3) NEO_LAUNCH_ROUTED(lds, 3, tame); else NEO_LAUNCH_ROUTED(lds, 2, tame); \
  } while (0)
#define NEO_LAUNCH_W(w, ...)                                                                             \
  do {                                                                                                   \
    if ((w) == 4) NEO_LAUNCH(4, __VA_ARGS__); else if ((w) == 3) NEO_LAUNCH(3, __VA_ARGS__); else NEO_LAUNCH(2, __VA_ARGS__); \
  } while (0)
#ifdef NEO_MPC_TU_RICCATI
  if (a.p.routed) {   // control_steps 3, AUTO: direction by neighbourhood (k_solve_routed; the dense layout, 4 waves/SIMD like the dense kernels)
    const int w = solve_variant(tuning, 4);
    if (disc && w == 4 && small_tile) NEO_LAUNCH_ROUTED(0, 4, true, 1024);   // (the static variant takes no dynamic LDS)
    else if (disc) NEO_LAUNCH_ROUTED_W(w, true);
    else NEO_LAUNCH_ROUTED_W(w, false);
  } else
  {  // any control_steps: Newton direction by the Riccati sweep (riccati.h)
    // the 128-VGPR build (4 waves/SIMD) wherever LDS lets a CU hold more than 12 workgroups -- 13 need <= 12.3 KB
    // each -- else the 168-VGPR build (measured: control_steps 8, 16 workgroups/CU: +17 %; control_steps 32 at
    // 11.3 KB = 14 workgroups/CU: +9 %; with 12 workgroups/CU the 4-wave build's spills make it 4 % slower; the general
    // variant's 10 spilled VGPRs at 4 waves/SIMD cost nothing measurable: "turn" parameter set, 65 536 instances, same
    // box: 34.9 M solves/s against 30.6 M at 3 waves/SIMD, tools/ab_general.py)
    const int w = solve_variant(tuning, lds <= 12600 ? 4 : 3);
    if (disc) NEO_LAUNCH_W(w, 0, 2, true);
    else NEO_LAUNCH_W(w, 0, 2);
  }
  (void)generic; (void)small_tile;
#else
  if (a.p.newton == 2) { launch_solve_riccati(a, tuning, stream, ev_start, ev_stop); return; }
  if (a.p.n == 3 && a.p.newton == 1) {  // projected Newton, dense 9 x 9 system (its layout does not depend on lbfgs_memory)
    // (the general variant at 4 waves/SIMD spills 7 VGPRs -- 32 bytes of scratch -- and is still the faster one: measured
    // on the "cut" parameter set, same box, tools/ab_general.py: 0.123 ms against 0.133 ms per 4096 instances at 3
    // waves/SIMD -- 4096 waves are one residency round at 4 --, 55.9 M against 51.3 M solves/s at 65 536)
    const int w = solve_variant(tuning, 4);
    if (a.p.routed) { launch_solve_riccati(a, tuning, stream, ev_start, ev_stop); return; }   // AUTO: direction by neighbourhood (k_solve_routed)
    if (disc && w == 4 && small_tile) NEO_LAUNCH_LDS(0, 4, 3, 1, true, 1024);   // (the static variant takes no dynamic LDS)
    else if (disc) NEO_LAUNCH_W(w, 3, 1, true);
    else NEO_LAUNCH_W(w, 3, 1);
  } else if (a.p.n == 3 && a.p.newton == 0 && !generic) {
    NEO_LAUNCH_W(solve_variant(tuning, 3), 3);
  } else if (a.p.newton == 1) {  // control_steps <= kNewtonMaxSteps, dense system with run-time size
    // (a 24-entry row per lane: 158 VGPRs, 187 without the tame specialisation -- spill-free at 3 and 2 waves/SIMD)
    const int w = solve_variant(tuning, disc ? 3 : 2);
    if (disc && w == 3 && small_tile) NEO_LAUNCH_LDS(0, 3, 0, 1, true, 1024);
    else if (disc) NEO_LAUNCH_W(w, 0, 1, true);
    else NEO_LAUNCH_W(w, 0, 1);
  } else {  // projected L-BFGS, any control_steps
    const int w = solve_variant(tuning, 3);
    if (disc) NEO_LAUNCH_W(w, 0, 0, true);
    else NEO_LAUNCH_W(w, 0, 0);
  }
#endif  // NEO_MPC_TU_RICCATI
#undef NEO_LAUNCH_ROUTED_W
#undef NEO_LAUNCH_ROUTED
#undef NEO_LAUNCH_W
#undef NEO_LAUNCH
#undef NEO_LAUNCH_LDS
}
#ifndef NEO_MPC_TU_RICCATI
void launch_carrots(const CarrotArgs& a, void* stream) {
  if (a.b.count == 0) return;
  hipLaunchKernelGGL(k_carrot, dim3((unsigned)a.b.count), dim3(kLanes), 0, (hipStream_t)stream, a);
}
void launch_postprocess(const SolveArgs& a, void* stream) {
  if (a.count == 0) return;
  hipLaunchKernelGGL(k_postprocess, dim3(a.count), dim3(kLanes), a.lds.total_bytes, (hipStream_t)stream, a);
}
void launch_objective(const ObjectiveArgs& a, void* stream) {
  if (a.count == 0) return;
  hipLaunchKernelGGL(k_objective, dim3((a.count + 255) / 256), dim3(256), 0, (hipStream_t)stream, a);
}
// K5.  A launch of up to 4096 instances is one residency round: workgroups w, w + 1024, w + 2048 and w + 3072 share a SIMD
// (measured with the XCC id in the hardware key) and the SIMD with the largest sum of iterations ends the
// launch -- 23.5 iteration-slots against a mean of 13.9 in the closed loop of 4096 robots.  Robots keep their habits from one
// tick to the next (correlation of the iteration counts of consecutive ticks 0.79: stopped robots 2, cruising robots 3,
// robots along a wall 8 and more), so past counts predict this tick's load: instances sorted by them, longest first, and
// dealt over the 1024 SIMDs in snake order (slot 0 left to right, slot 1 right to left, ...) bring the maximum down to 21
// with the last tick's counts alone, to 20.3 with `load` = an exponential average over the calls (decay 1/2: the handle
// keeps it between calls; on the mirror's closed loop, the order rebuilt every 5th tick).  One workgroup, a counting sort
// over 512 bins of a quarter of an iteration; ranks among equals come from atomics (any order of equals is as good as
// another).  Counts that are not a multiple of 1024 or beyond 4096 (more than one round: the hardware deals waves as slots
// free up) get the identity.
__global__ __launch_bounds__(1024) void k_dispatch_order(const neo_mpc_command* commands, float* load, uint32_t* order,
                                                         uint32_t count, int fresh) {
  constexpr uint32_t kBins = 512;
  __shared__ uint32_t hist[kBins], scan[kBins];
  const uint32_t t = threadIdx.x;
  if (count % kDispatchSimds != 0 || count > 4 * kDispatchSimds) {
    for (uint32_t i = t; i < count; i += 1024) order[i] = i;
    return;
  }
  if (t < kBins) hist[t] = 0;
  __syncthreads();
  uint32_t key[4], rank[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const uint32_t i = t + 1024u * k;
    if (i < count) {
      const int it = commands[i].iterations;
      const float now = (float)(it < 0 ? 0 : it > 127 ? 127 : it);
      const float e = fresh ? 2.0f * now : 0.5f * load[i] + now;     // (a fresh average starts at its steady state)
      load[i] = e;
      const uint32_t q = (uint32_t)fminf(2.0f * e + 0.5f, (float)(kBins - 1));   // (e is twice the count in the steady state)
      key[k] = kBins - 1u - q;                                        // ascending key = descending load
      rank[k] = atomicAdd(&hist[key[k]], 1u);
    }
  }
  __syncthreads();
  // exclusive prefix sum over the bins (Hillis-Steele, 9 steps, two buffers)
  uint32_t* src = hist;
  uint32_t* dst = scan;
  for (uint32_t d = 1; d < kBins; d <<= 1) {
    if (t < kBins) dst[t] = src[t] + (t >= d ? src[t - d] : 0u);
    __syncthreads();
    uint32_t* tmp = src; src = dst; dst = tmp;
  }
  // (src holds the inclusive sums)
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const uint32_t i = t + 1024u * k;
    if (i < count) {
      const uint32_t pos = (key[k] ? src[key[k] - 1u] : 0u) + rank[k], slot = pos / kDispatchSimds;
      uint32_t lane = pos % kDispatchSimds;
      if (slot & 1u) lane = kDispatchSimds - 1u - lane;
      order[slot * kDispatchSimds + lane] = i;
    }
  }
}
void launch_dispatch_order(const neo_mpc_command* commands, float* load, uint32_t* order, uint32_t count, bool fresh, void* stream) {
  if (count == 0) return;
  hipLaunchKernelGGL(k_dispatch_order, dim3(1), dim3(1024), 0, (hipStream_t)stream, commands, load, order, count, fresh ? 1 : 0);
}
void launch_ingest(const IngestArgs& a, const LaunchTuning& tuning, void* stream) {
  const long total = (long)a.rows * (a.pitch >> 4);
  // a few 16-byte chunks per thread: one-chunk threads make the launch dispatch-bound for pools of small maps
  const int per_thread = kIngestUnroll;
  int blocks = (int)((total + 256L * per_thread - 1) / (256L * per_thread));
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(k_ingest, dim3(blocks, a.maps > 0 ? a.maps : 1), dim3(256), 0, (hipStream_t)stream, a);
}
#endif  // NEO_MPC_TU_RICCATI

}  // namespace neo_mpc
