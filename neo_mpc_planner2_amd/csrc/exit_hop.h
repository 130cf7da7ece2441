// exit_hop.h -- what a dense-direction search of K1 does when it has ended: one look for a cheaper costmap cell a hop away.
#pragma once
#include "neo_mpc_device.h"
#include "wave_ops.h"
#include "solver_context.h"
#include "costmap.h"
#include "feasible_set.h"
#include "rollout.h"

namespace neo_mpc {
namespace {

// Dense direction: a search that has ENDED looks once for a cheaper costmap cell a hop away -- the hop candidates
// the stage-wise direction tries in every iteration (costmap.h edge_stickiness).  A search that closed in on a cell edge
// from the expensive side ends a millimetre short of a cost step no descent direction sees (held-out parameter set "a",
// w_costmap / w_trans = 0.08: one such step is worth 2e-3, P3 failed on 1 of 24 cases).  A hop that lowers the objective is
// taken; the search is not taken up again (measured on the mirror: restarting it gains 7e-6 per instance on average
// and lengthens the slowest searches of a launch by two iterations).  Skipped when no stage of the iterate has a costmap
// term under it: nothing is cheaper next door.
// In: the final iterate u (LDS) and its objective f.  Out: u with ONE block changed and the lower f, or both as they were;
// nfev counts the evaluation.  (The caller skips it when no stage of the iterate has a costmap term under it, or the
// whole reach tile is free.)
template <int kSteps, int kNwSteps, bool kTame, bool kCovered>
__device__ __forceinline__ void exit_hop(const SolveArgs& a, const Ctx& c, double* L, double& f, int& nfev, int lane, int n) {
  const DevParams& p = a.p;
  double* u = L + a.lds.u;
  bool has_hop = false;
  float hop_x = 0.0f, hop_y = 0.0f;
  {
    // lane i < n: position and heading of stage i at u
    double x = 0.0, y = 0.0, th = 0.0, cs = 1.0, sn = 0.0;
#pragma unroll
    for (int k = 0; k < kNwSteps; ++k) {
      if ((kSteps || k < n) && k <= lane) {
        th += u[3 * k + 2] * p.dt;
        sincos_heading<kTame>(th, &sn, &cs);
        x += (u[3 * k] * cs - u[3 * k + 1] * sn) * p.dt;
        y += (u[3 * k] * sn + u[3 * k + 1] * cs) * p.dt;
      }
    }
    if (lane < n) {
      double wxx, wxy, wyy, wlx, wly;
      (void)edge_stickiness<kCovered>(a, c, L, x, y, cs, sn, wxx, wxy, wyy, wlx, wly, has_hop, hop_x, hop_y);
    }
  }
  const unsigned long long hmask = __ballot(has_hop);
  if (hmask != 0ull) {   // (wave-uniform)
    const int rank = __popcll(hmask & ((1ull << lane) - 1ull));
    double* t = L + a.lds.tol;
    if (has_hop && rank < kHopLanes) {
      reinterpret_cast<int*>(t + T_HOP_STAGE)[rank] = lane;
      reinterpret_cast<float*>(t + T_HOP_VEC)[2 * rank] = hop_x;
      reinterpret_cast<float*>(t + T_HOP_VEC)[2 * rank + 1] = hop_y;
    }
    const int nh = min(__popcll(hmask), (int)kHopLanes);
    WAVE_SYNC();
    // lane h < nh: the current point with the block of hop stage h changed
    int hs = -1;
    float hx = 0.0f, hy = 0.0f;
    if (lane < nh) {
      hs = reinterpret_cast<const int*>(t + T_HOP_STAGE)[lane];
      hx = reinterpret_cast<const float*>(t + T_HOP_VEC)[2 * lane];
      hy = reinterpret_cast<const float*>(t + T_HOP_VEC)[2 * lane + 1];
    }
    double hb0 = 0.0, hb1 = 0.0;
    double fh = rollout_cost<kSteps, kTame, kCovered>(
        a, c, L,
        [&](int i, double& b0, double& b1, double& b2) {
          b0 = u[3 * i]; b1 = u[3 * i + 1]; b2 = u[3 * i + 2];
          if (i == hs) { b0 += (double)hx; b1 += (double)hy; project_block<kTame>(p, b0, b1, b2); hb0 = b0; hb1 = b1; }
        });
    if (!(fh == fh) || lane >= nh) fh = INFINITY;
    int hbest = lane;
    wave_argmin(fh, hbest);
    ++nfev;
    WAVE_SYNC();
    if (fh < f) {
      if (lane == hbest) { u[3 * hs] = hb0; u[3 * hs + 1] = hb1; }
      f = fh;
      WAVE_SYNC();
    }
  }
}

}  // namespace
}  // namespace neo_mpc
