// tangent_cone.h -- the active-set pass of K1's solver loop: total gradient, tangent-cone reduction, face records.
//
// For every control block (lane = block): the total gradient gt = smooth gradient + gradient of the control norm (the
// minimal-norm subgradient ON the kink); the constraints active at u_i (vx / vy bounds, the max_vel_trans disc, the omega
// bounds) and which of them the descent direction pushes into; the reduced gradient gr = minus the projection of -gt onto
// the tangent cone; the face the second-order directions work on (mode 0 free / 1 sliding along one constraint / 2 pinned,
// outward normal, omega frozen) and its records -- projector + block curvature for the dense Newton system, disc
// curvature for the stage-wise sweep.  Blocks within kink_radius of the kink u_i = v_cur are "near": moved by the
// proximal step only, their SMOOTH gradient reduced on the cone and written back to gs.
// LDS in: u, gs.  LDS out: gt, gr, (gs of near blocks), nx, ny, mode[4 i .. 4 i + 2], Newton records / disc curvature.
// Returns whether a block of this lane slides along a constraint of a feasible set that has corners (repin_corner_blocks).
#pragma once
#include "neo_mpc_device.h"
#include "fast_math.h"
#include "solver_context.h"
#include "solver_rules.h"
#include "riccati.h"

namespace neo_mpc {
namespace {

constexpr int kNewtonRecord = 13;   // float32 record per control block of the dense Newton system: P00 P01 P11 PW, C (3 x 3)

template <bool kTame, bool kNewton, bool kRiccati>
__device__ __forceinline__ bool tangent_cone_pass(const SolveArgs& a, const Ctx& c, double* L, double kink_radius, int lane, int n) {
  constexpr bool kSecond = kNewton || kRiccati;
  const DevParams& p = a.p;
  const double* u = L + a.lds.u;
  double* gs = L + a.lds.gs;
  double* gt = L + a.lds.gt;
  double* gr = L + a.lds.gr;
  double* ANX = L + a.lds.nx;
  double* ANY = L + a.lds.ny;
  int* AMODE = reinterpret_cast<int*>(L + a.lds.mode);
  float* NB = reinterpret_cast<float*>(L + a.lds.cs);    // (dense Newton: the records live in the step arrays it does not use)
  float* ARTF = reinterpret_cast<float*>(L + a.lds.rt);  // (stage-wise: [2 i] disc curvature lambda / r of block i)
  bool my_corner = false;   // a block of this lane slides along one constraint in a corner of the feasible set
  for (int i = lane; i < n; i += kLanes) {
    const double u0 = u[3 * i], u1 = u[3 * i + 1], u2 = u[3 * i + 2];
    const double g0 = gs[3 * i], g1 = gs[3 * i + 1], g2 = gs[3 * i + 2];
    const double e0 = u0 - c.v0, e1 = u1 - c.v1, e2 = u2 - c.v2;
    // |e| and 1/|e| from one reciprocal square root (a shorter dependent chain than sqrt, then rcp)
    const double ne2 = e0 * e0 + e1 * e1 + e2 * e2;
    const double ine = ne2 > 0.0 ? rsq_fast(ne2) : 0.0;
    const double ne = ne2 * ine;
    double t0, t1, t2;
    if (ne2 > 0.0) {
      const double wn = p.wc_n * ine;
      t0 = g0 + wn * e0; t1 = g1 + wn * e1; t2 = g2 + wn * e2;
    } else {
      const double ng2 = g0 * g0 + g1 * g1 + g2 * g2;
      const double sh = (ng2 > p.wc_n * p.wc_n) ? 1.0 - p.wc_n * rsq_fast(ng2) : 0.0;
      t0 = g0 * sh; t1 = g1 * sh; t2 = g2 * sh;
    }
    if (!kRiccati) AMODE[4 * i + 3] = AMODE[4 * i + 2];   // (Riccati: slot 3 is its to-the-kink flag)
    // next to the kink: prox-only block, outside the quasi-Newton model.  Its SMOOTH gradient is reduced on the tangent
    // cone below (same code path as every other block's total gradient) and written back to gs: the proximal step of
    // such a block is taken on its face -- a component that pushes omega into its bound, or the velocity out of the
    // disc, used to dominate the step's shrink factor and keep the block from landing on the kink (projection and prox
    // do not commute).
    const bool near = ne < kink_radius;
    if (near) { t0 = g0; t1 = g1; t2 = g2; }
    else { gt[3 * i] = t0; gt[3 * i + 1] = t1; gt[3 * i + 2] = t2; }
    const int wfroz = ((u2 <= p.lo[2] && t2 > 0.0) || (u2 >= p.hi[2] && t2 < 0.0)) ? 1 : 0;
    double r0 = t0, r1 = t1;
    // outward normals of the constraints active at u: slot 0 = vx bound, 1 = vy bound, 2 = disc
    // (fixed slots + validity flags: no dynamically indexed private arrays, i.e. no scratch)
    double nx0 = 0.0, ny0 = 0.0, nx1 = 0.0, ny1 = 0.0, nx2 = 0.0, ny2 = 0.0;
    bool v0 = false, v1 = false, v2 = false;
    if (!kTame && !p.disc_in_box) {  // (inside the box a bound can only touch where the disc touches too)
      // (active within NEO_RULE_CORNER_ROOM: the projection's radial rescale leaves a block that sat on a bound a rounding
      // error inside it -- seen as free, a block in the corner between the disc and a bound slid along the disc INTO the
      // bound, was pinned by the corner re-pin and never tried the slide along the bound: round-5 review, seed 62022)
      constexpr double kRoom = NEO_RULE_CORNER_ROOM;
      if (u0 <= p.lo[0] + kRoom) { nx0 = -1.0; v0 = true; }
      else if (u0 >= p.hi[0] - kRoom) { nx0 = 1.0; v0 = true; }
      if (u1 <= p.lo[1] + kRoom) { ny1 = -1.0; v1 = true; }
      else if (u1 >= p.hi[1] - kRoom) { ny1 = 1.0; v1 = true; }
    }
    const double nvv2 = u0 * u0 + u1 * u1, rlim = p.r * (1.0 - 1e-12);
    if (nvv2 > 0.0 && nvv2 >= rlim * rlim) { const double iv = rsq_fast(nvv2); nx2 = u0 * iv; ny2 = u1 * iv; v2 = true; }
    const double dx = -t0, dy = -t1;
    const double dn0 = nx0 * dx + ny0 * dy, dn1 = nx1 * dx + ny1 * dy, dn2 = nx2 * dx + ny2 * dy;
    int mode = 0, mslot = -1;
    double mnx = 0.0, mny = 0.0, mlam = 0.0;
    if ((v0 && dn0 > 0.0) || (v1 && dn1 > 0.0) || (v2 && dn2 > 0.0)) {
      // slide along one violated constraint if that keeps the others satisfied; longest slide wins
      double bestn = -1.0;
      mode = 2;
#define NEO_TRY_SLIDE(sk, vk, dnk, nxk, nyk, va, nxa, nya, vb, nxb, nyb)                               \
      if (vk && dnk > 0.0) {                                                                         \
        const double px = dx - dnk * nxk, py = dy - dnk * nyk;                                       \
        const double tol = 1e-14 * (fabs(px) + fabs(py));                                            \
        const bool ok = !(va && nxa * px + nya * py > tol) && !(vb && nxb * px + nyb * py > tol);    \
        const double pn = px * px + py * py;                                                         \
        if (ok && pn > bestn) {                                                                      \
          bestn = pn; mode = 1; mnx = nxk; mny = nyk; r0 = -px; r1 = -py; mslot = sk; mlam = dnk;     \
        }                                                                                            \
      }
      // (the disc first: where a bound touches the disc with the same normal -- max_vel_x = max_vel_trans, README -- the
      // slide is the disc's, with its curvature and without the corner stop of a bound slide)
      NEO_TRY_SLIDE(2, v2, dn2, nx2, ny2, v0, nx0, ny0, v1, nx1, ny1)
      NEO_TRY_SLIDE(0, v0, dn0, nx0, ny0, v1, nx1, ny1, v2, nx2, ny2)
      NEO_TRY_SLIDE(1, v1, dn1, nx1, ny1, v0, nx0, ny0, v2, nx2, ny2)
#undef NEO_TRY_SLIDE
      if (mode == 2) { r0 = 0.0; r1 = 0.0; }
    }
    if (near) {
      const double s2 = wfroz ? 0.0 : t2;
      gs[3 * i] = r0; gs[3 * i + 1] = r1; gs[3 * i + 2] = s2;
      gt[3 * i] = 0.0; gt[3 * i + 1] = 0.0; gt[3 * i + 2] = 0.0;
      gr[3 * i] = 0.0; gr[3 * i + 1] = 0.0; gr[3 * i + 2] = 0.0;
      // (2: exactly ON the kink with a REDUCED smooth gradient inside the norm's subdifferential, |g_s| <= w_control/N --
      // the block stays there under every proximal step: at rest, it does not hold up the Newton stop tests)
      const int at_rest = (ne2 == 0.0 && r0 * r0 + r1 * r1 + s2 * s2 <= p.wc_n * p.wc_n) ? 2 : 1;
      ANX[i] = 0.0; ANY[i] = 0.0; AMODE[4 * i] = 0; AMODE[4 * i + 1] = 0; AMODE[4 * i + 2] = at_rest;
      if (kNewton) {
#pragma unroll
        for (int k = 0; k < kNewtonRecord; ++k) NB[kNewtonRecord * i + k] = 0.0f;  // P = 0: row/column of I
      }
      if (kRiccati) ARTF[2 * i] = 0.0f;
      continue;
    }
    gr[3 * i] = r0; gr[3 * i + 1] = r1; gr[3 * i + 2] = wfroz ? 0.0 : t2;
    // (slot 1: omega frozen | 2 x "the slide is along the disc" -- a block sliding along a box bound stops at the corner
    // where the bound meets the disc, feasible_set.h candidate_block)
    // (second-order directions, box cutting the disc: a sliding block with no room left towards ANOTHER constraint sits in
    // a corner of the feasible set -- repin_corner_blocks looks at where the direction sends it)
    if (kSecond && !kTame && !p.disc_in_box && mode == 1) {
      const double room = NEO_RULE_CORNER_ROOM, rin = p.r - room;
      my_corner = my_corner || (mslot != 0 && (u0 - p.lo[0] <= room || p.hi[0] - u0 <= room)) ||
                  (mslot != 1 && (u1 - p.lo[1] <= room || p.hi[1] - u1 <= room)) || (mslot != 2 && nvv2 >= rin * rin);
    }
    ANX[i] = mnx; ANY[i] = mny; AMODE[4 * i] = mode; AMODE[4 * i + 1] = wfroz | ((mode == 1 && mslot == 2) ? 2 : 0); AMODE[4 * i + 2] = 0;
    if (kNewton) {
      // Block record of the Newton system, float32: the projector onto the tangent cone's face
      // (P00 P01 P11 PW) and the block's own curvature C (3x3): the control norm's Hessian
      // (w/|e|)(I - e e^T/|e|^2) plus lambda/r t t^T of a binding disc, t = (-ny, nx)
      float* nb = NB + kNewtonRecord * i;
      nb[0] = mode == 0 ? 1.0f : mode == 1 ? (float)(1.0 - mnx * mnx) : 0.0f;
      nb[1] = mode == 1 ? (float)(-mnx * mny) : 0.0f;
      nb[2] = mode == 0 ? 1.0f : mode == 1 ? (float)(1.0 - mny * mny) : 0.0f;
      nb[3] = wfroz ? 0.0f : 1.0f;
      const float f0 = (float)e0, f1 = (float)e1, f2 = (float)e2;
      const float fn2 = f0 * f0 + f1 * f1 + f2 * f2;
      const float ine = fn2 > 0.0f ? __builtin_amdgcn_rsqf(fn2) : 0.0f;
      const float sN = (float)p.wc_n * ine, h0 = f0 * ine, h1 = f1 * ine, h2 = f2 * ine;
      const float k2 = (mode == 1 && mslot == 2) ? (float)(mlam * rcp_fast(p.r)) : 0.0f;
      const float tx = -(float)mny, ty = (float)mnx;
      const float c00 = sN * (1.0f - h0 * h0) + k2 * tx * tx, c01 = -sN * h0 * h1 + k2 * tx * ty,
                  c02 = -sN * h0 * h2, c11 = sN * (1.0f - h1 * h1) + k2 * ty * ty, c12 = -sN * h1 * h2,
                  c22 = sN * (1.0f - h2 * h2);
      nb[4] = c00; nb[5] = c01; nb[6] = c02;
      nb[7] = c01; nb[8] = c11; nb[9] = c12;
      nb[10] = c02; nb[11] = c12; nb[12] = c22;
    }
    // Riccati: curvature lambda/r of a binding disc (the rest of the block's record is made by riccati_prepare)
    if (kRiccati) ARTF[2 * i] = (mode == 1 && mslot == 2) ? (float)(mlam * rcp_fast(p.r)) : 0.0f;
  }
  return my_corner;
}

// One-sided slides (second-order directions, round 4).  A block in a CORNER of the feasible set -- two constraints active
// -- that slides along one of them can only slide away from the other.  A Newton step that sends it the other way is
// stopped by the projection of every candidate while all the other blocks take the step that counted on it: the
// direction is then no descent direction at any length (held-out fuzz, box cutting the disc at control_steps 10: a search
// that crept for 20 iterations on proximal steps alone and stopped 1.4e-2 short).  Such blocks are pinned (mode 2, reduced
// gradient zero, projector / stage case of a pinned block) and the caller computes the direction once more: one round of
// a QP solver's active-set step.  Returns whether any block was pinned (wave-uniform).
template <bool kRiccati>
__device__ __forceinline__ bool repin_corner_blocks(const SolveArgs& a, double* L, int n, int lane) {
  int* AMODE = reinterpret_cast<int*>(L + a.lds.mode);
  const double* d = L + a.lds.d;
  const double* u = L + a.lds.u;
  double* gr = L + a.lds.gr;
  const DevParams& p = a.p;
  bool any = false;
  for (int i = lane; i < n; i += kLanes) {
    int* am = AMODE + 4 * i;
    if (am[0] != 1 || (kRiccati && am[3])) continue;
    // the OTHER constraints of the block (it slides along the disc, or along the vx or the vy bound): does the step approach
    // one of them with no room left (NEO_RULE_CORNER_ROOM: the projection's rounding leaves a block 1e-16 inside a bound
    // it sat on)?
    const double u0 = u[3 * i], u1 = u[3 * i + 1], d0 = d[3 * i], d1 = d[3 * i + 1];
    const bool on_disc = (am[1] & 2) != 0, on_x = !on_disc && L[a.lds.nx + i] != 0.0, on_y = !on_disc && !on_x;
    bool blocked = false;
    if (!on_x) blocked = blocked || (d0 != 0.0 && (d0 > 0.0 ? p.hi[0] - u0 : u0 - p.lo[0]) <= NEO_RULE_CORNER_ROOM);
    if (!on_y) blocked = blocked || (d1 != 0.0 && (d1 > 0.0 ? p.hi[1] - u1 : u1 - p.lo[1]) <= NEO_RULE_CORNER_ROOM);
    if (!on_disc) {
      const double rin = p.r - NEO_RULE_CORNER_ROOM;
      blocked = blocked || (d0 * u0 + d1 * u1 > 0.0 && u0 * u0 + u1 * u1 >= rin * rin);
    }
    if (!blocked) continue;
    any = true;
    am[0] = 2; gr[3 * i] = 0.0; gr[3 * i + 1] = 0.0;
    if (kRiccati) {
      float* rs = reinterpret_cast<float*>(L + a.lds.ric) + kRicStage * i;
      const int flags = (int)rs[RS_FLAGS], kase = flags & RF_CASE;
      rs[RS_FLAGS] = (float)((flags & ~RF_CASE) | (kase == RC_SLIDE_W ? RC_W : kase == RC_SLIDE ? RC_NONE : kase));
    } else {
      float* nb = reinterpret_cast<float*>(L + a.lds.cs) + kNewtonRecord * i;
      nb[0] = 0.0f; nb[1] = 0.0f; nb[2] = 0.0f;
    }
  }
  const bool redo = __ballot(any) != 0ull;
  WAVE_SYNC();
  return redo;
}

}  // namespace
}  // namespace neo_mpc
