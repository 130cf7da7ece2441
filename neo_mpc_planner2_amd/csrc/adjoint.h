// adjoint.h -- the gradient passes of K1's solver loop: rollout + adjoint sweep of the tracking and terminal cost
// (py:230-266 differentiated by hand; the control norm and the costmap term are dealt with elsewhere: tangent_cone.h,
// costmap.h).  Shapes:
//   (dense Newton: in k_solve itself -- every lane runs the sweep on its own copy of u, lane k < 3N perturbed in coordinate
//    k; one pass yields the gradient and all 3N Hessian columns, which stay in that function's registers)
//   adjoint_by_scans              any control_steps <= 64 (L-BFGS, stage-wise): lane = stage, the rollout recursion is three
//                                 prefix sums and the adjoint three suffix sums; the stage-wise direction's records, wall
//                                 model and hop table are filled on the way
//   adjoint_short_sweep           control_steps specialisations of L-BFGS: all lanes walk the same short sweep
// LDS in: u.  LDS out: gs (smooth gradient) and what each shape says above.
#pragma once
#include "neo_mpc_device.h"
#include "wave_ops.h"
#include "fast_math.h"
#include "solver_context.h"
#include "costmap.h"
#include "riccati.h"

namespace neo_mpc {
namespace {

// exact_step: the stage records carry the position costates (second-order terms of the rollout step, riccati.h).
// free_path: every stage of the rollout at u sits in a free cell; nhops: hop candidates in the tolerance block's table.
// kFew > 0: control_steps == kFew at compile time (at most 16): the prefix sums stop after the DPP steps that reach the stages
template <bool kTame, bool kRiccati, int kFew = 0>
__device__ __forceinline__ void adjoint_by_scans(const SolveArgs& a, const Ctx& c, double* L, bool exact_step, int lane, int n,
                                                 bool& free_path, int& nhops) {
  auto scan = [](double v) { if constexpr (kFew > 0) return wave_scan_few<kFew>(v); else return wave_scan(v); };
  constexpr int kLast = kFew > 0 ? kFew - 1 : 63;   // (the lane that holds a prefix sum's total)
  const DevParams& p = a.p;
  const double* u = L + a.lds.u;
  double* gs = L + a.lds.gs;
  float* RS = reinterpret_cast<float*>(L + a.lds.ric);
  float* ARTF = reinterpret_cast<float*>(L + a.lds.rt);
  // (L-BFGS and Riccati) any control_steps <= 64: lane i owns step i; the rollout recursion (py:230-232) is three
  // prefix sums, the adjoint three suffix sums -- one sincos per lane instead of N in a row
  const bool on = lane < n;
  const double vx = on ? u[3 * lane] : 0.0, vy = on ? u[3 * lane + 1] : 0.0, w = on ? u[3 * lane + 2] : 0.0;
  const double th = scan(w * p.dt);
  double sn, cs;
  sincos_heading<kTame>(th, &sn, &cs);
  const double ddx = (vx * cs - vy * sn) * p.dt, ddy = (vx * sn + vy * cs) * p.dt;
  const double x = scan(ddx), y = scan(ddy);
  double rt = on ? -2.0 * p.wo_n * (c.tyaw - th) : 0.0;
  if (lane == n - 1) rt += -2.0 * p.wterm_o * (c.fyaw - th);
  const double rx = on ? -2.0 * p.wt_n * (c.cx - x) : 0.0, ry = on ? -2.0 * p.wt_n * (c.cy - y) : 0.0;
  // suffix sums: S_k = sum_{i >= k} r_i = total - prefix_k + r_k
  const double px = scan(rx), py = scan(ry);
  const double SX = lane_value(px, kLast) - px + rx, SY = lane_value(py, kLast) - py + ry;
  const double tt = on ? rt - ddy * SX + ddx * SY : 0.0;
  const double pt = scan(tt);
  const double ST = lane_value(pt, kLast) - pt + tt;
  int raw_here = 0;
  bool has_hop = false;
  float hop_x = 0.0f, hop_y = 0.0f;
  if (on) {
    gs[3 * lane] = p.dt * (cs * SX + sn * SY);
    gs[3 * lane + 1] = p.dt * (-sn * SX + cs * SY);
    gs[3 * lane + 2] = p.dt * ST;
    if (kRiccati) {
      // stage record of the Riccati sweep (float32): trigonometry, position increments and the
      // wall-sliding penalty on the stage position (costmap.h)
      float* rs = RS + kRicStage * lane;
      rs[RS_CS] = (float)cs; rs[RS_SN] = (float)sn; rs[RS_PX] = (float)ddx; rs[RS_PY] = (float)ddy;
      double wxx, wxy, wyy, wlx, wly;
      raw_here = edge_stickiness(a, c, L, x, y, cs, sn, wxx, wxy, wyy, wlx, wly, has_hop, hop_x, hop_y);
      rs[RS_WXX] = (float)wxx; rs[RS_WXY] = (float)wxy; rs[RS_WYY] = (float)wyy;
      rs[RS_WLX] = (float)wlx; rs[RS_WLY] = (float)wly;
      // position costates of this stage, for the second-order terms of the rollout step (riccati.h): only behind an
      // iteration won by a decent Newton step (the model held there) -- far from the minimiser the exact Hessian is
      // indefinite and the Gauss-Newton direction is the safer one
      rs[RS_SY] = exact_step ? (float)SY : 0.0f;
      ARTF[2 * lane + 1] = exact_step ? (float)SX : 0.0f;
    }
  }
  if (kRiccati) {
    free_path = __ballot(raw_here != 0) == 0ull;
    // hop table of this iteration: the first kHopLanes stages with a cheaper cell a hop away (lanes 1.. of the search)
    const unsigned long long hmask = __ballot(has_hop);
    const int rank = __popcll(hmask & ((1ull << lane) - 1ull));
    double* t = L + a.lds.tol;
    if (has_hop && rank < kHopLanes) {
      reinterpret_cast<int*>(t + T_HOP_STAGE)[rank] = lane;
      reinterpret_cast<float*>(t + T_HOP_VEC)[2 * rank] = hop_x;
      reinterpret_cast<float*>(t + T_HOP_VEC)[2 * rank + 1] = hop_y;
    }
    nhops = min(__popcll(hmask), (int)kHopLanes);
    if (lane == 0) reinterpret_cast<int*>(t + T_HOP_STAGE)[kHopLanes] = nhops;
  }
  WAVE_SYNC();
}

// have_trig: the step arrays already hold sin / cos of the rollout at u (the winner's, stored by the candidate pass)
template <int kSteps, bool kTame>
__device__ __forceinline__ void adjoint_short_sweep(const SolveArgs& a, const Ctx& c, double* L, bool have_trig, int lane, int n) {
  const DevParams& p = a.p;
  const double* u = L + a.lds.u;
  double* gs = L + a.lds.gs;
  double* ACS = L + a.lds.cs;
  double* ASN = L + a.lds.sn;
  double* ADX = L + a.lds.dxs;
  double* ADY = L + a.lds.dys;
  double* ARX = L + a.lds.rx;
  double* ARY = L + a.lds.ry;
  double* ART = L + a.lds.rt;
  // specialisations: all lanes walk the same short sweep, reusing the winner's sin/cos
  double x = 0.0, y = 0.0, th = 0.0;
#pragma unroll
  for (int i = 0; i < n; ++i) {
    const double vx = u[3 * i], vy = u[3 * i + 1], w = u[3 * i + 2];
    th += w * p.dt;
    double sn, cs;
    if (kSteps && have_trig) { sn = ASN[i]; cs = ACS[i]; }
    else sincos_heading<kTame>(th, &sn, &cs);
    const double ddx = (vx * cs - vy * sn) * p.dt, ddy = (vx * sn + vy * cs) * p.dt;
    x += ddx; y += ddy;
    double rt = -2.0 * p.wo_n * (c.tyaw - th);
    if (i == n - 1) rt += -2.0 * p.wterm_o * (c.fyaw - th);
    if (lane == 0) {
      ACS[i] = cs; ASN[i] = sn; ADX[i] = ddx; ADY[i] = ddy;
      ARX[i] = -2.0 * p.wt_n * (c.cx - x); ARY[i] = -2.0 * p.wt_n * (c.cy - y); ART[i] = rt;
    }
  }
  WAVE_SYNC();
  double SX = 0.0, SY = 0.0, ST = 0.0;
#pragma unroll
  for (int k = n - 1; k >= 0; --k) {
    SX += ARX[k]; SY += ARY[k];
    ST += ART[k] - ADY[k] * SX + ADX[k] * SY;
    if (lane == 0) {
      gs[3 * k] = p.dt * (ACS[k] * SX + ASN[k] * SY);
      gs[3 * k + 1] = p.dt * (-ASN[k] * SX + ACS[k] * SY);
      gs[3 * k + 2] = p.dt * ST;
    }
  }
  WAVE_SYNC();
}

}  // namespace
}  // namespace neo_mpc
