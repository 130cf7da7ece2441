// neo_mpc_device.h -- host/device shared argument blocks of the gfx950 kernels.
// Internal to libneo_mpc.so (the public surface is include/neo_mpc.h).
#pragma once
#include <stdint.h>

#include "../../include/neo_mpc.h"
#include "solver_rules.h"

namespace neo_mpc {

constexpr int kLanes = 64;          // one CDNA4 wavefront per MPC instance
constexpr int kMapBorder = 64;      // lethal border (cells) K3 adds around a single costmap
constexpr int kPoolBorder = 16;     // ... around each map of a pool (every lookup is bounds-checked anyway)
constexpr int kMaxTileWidth = NEO_RULE_MAX_TILE_WIDTH;  // widest reach tile staged in LDS (bytes per row; solver_rules.h)
constexpr int kCompatNoUnshift = 0x10000;   // DevParams.compat (internal bit): searches always start at the warm start (NEO_MPC_COMPAT_REFERENCE_START)
constexpr int kDumpGradient = 0x40000000;  // DevParams.max_it value of the gradient test hook (neo_mpc_gradient_batch)

// Constants of one solver configuration, precomputed on the host in float64 exactly as
// the reference evaluates them (mpc_optimization_server.py:137, 252-268).
struct DevParams {
  double dt;                 // prediction_horizon / control_steps (py:137)
  double wt_n, wo_n, wc_n;   // w_trans/N, w_orient/N, w_control/N (py:252-254)
  double wterm_o;            // w_terminal * w_orient (py:268)
  double wterm_t;            // w_terminal * w_trans  (py:268, constant in u)
  double w_footprint;        // py:263 (constant in u)
  double lo[3], hi[3];       // box bounds vx, vy, omega (py:127-133)
  double r;                  // max_vel_trans (py:158)
  double acc[3];             // post clamp (py:385-391)
  double low_pass_gain;      // py:367
  double xtol;               // step tolerance
  double early_tol;          // Newton: stop when the full step is below this (= xtol; 0 disables, A/B)
  double final_tol;          // Newton: a full step below this is the last one (= opt_tolerance)
  double ftol;               // relative cost decrease below which an iteration counts as stalled
  double stall_step;         // ... or max|du| below this
  double wtol;               // three iterations in a row gaining less than this (relative) end the search; 0: off
  double wtol_late;          // ... the same from iteration kLateIteration on (the control_steps-3 window)
  double kink_radius;        // |u_i - v_cur| below which a block is handled by the prox step only
  double kink_radius_stagewise;  // ... for the instances the routed kernel sends to the stage-wise direction
  double btol_map, btol_free;  // dense Newton direction: kBlockedRun consecutive iterations not won by a decent Newton step that
                             // together gain less than this end the search (with / without a costmap term under the rollout); 0: off
  double hop_min_drop;       // stage-wise direction: a cheaper neighbour cell is worth a hop candidate when its costmap
                             // term is lower by more than this (0.1 * opt_tolerance)
  double hop_range;          // ... and its edge is closer than this (cells): min(0.25, 0.05 m/s * dt / resolution)
  double scan_resume_gain;   // cell scan (cell_scan.h): the search is taken up again behind a scan that gained more than this
  int32_t n;                 // control_steps
  int32_t max_it;
  int32_t mem;               // L-BFGS pairs
  int32_t compat;
  int32_t disc_in_box;       // the max_vel_trans disc lies inside the vx/vy box (README params)
  int32_t tame;              // disc_in_box and max|omega| * horizon <= 0.78 rad: the kTame kernels apply
  int32_t scan_reach;        // cell scan: no cell further than this from the robot's own cell (= the reach tile's radius)
  int32_t newton;            // search direction of lanes 32-63: 0 projected L-BFGS, 1 projected Newton with
                             // the dense system (control_steps <= 8), 2 projected Newton by the Riccati sweep
  int32_t routed;            // direction by neighbourhood (AUTO at control_steps 3, solver_rules.h): `newton` is the direction of
                             // the instances whose reach tile is all free, the others take the stage-wise one (k_solve_routed)
};

// Device costmap written by the ingest kernel (K3): raw nav2 costs with a lethal border of
// kMapBorder cells and a 128-byte row pitch; `cells` points at cell (0, 0).
struct DevMap {
  const uint8_t* cells;
  int32_t size_x, size_y;
  int32_t pitch;
  double resolution, inv_resolution, origin_x, origin_y;
  // costmap pool (neo_mpc_set_costmap_pool): pool_count bordered maps of this geometry back to back,
  // pool_stride bytes apart, their origins in pool_origins[2k], [2k+1]; 0: the single map above
  int32_t pool_count;
  int32_t border;    // lethal cells around the map in device memory (multiple of 16)
  int64_t pool_stride;
  const double* pool_origins;
};

// LDS carve-up of the solve kernel, in doubles from the start of dynamic LDS.
struct LdsLayout {
  int32_t prob, state, tol, term, u, gs, gt, gr, d, u_prev, gt_prev, u_new, S, Y, rho;
  int32_t cs, sn, dxs, dys, rx, ry, rt, nx, ny, mode;  // per-step scratch
  int32_t hess;        // Newton: (3N)^2 Hessian, only when 3N <= 24
  int32_t ric;         // Riccati: float32 stage records (28 N floats; the gains overwrite part of them), riccati.h
  int32_t keep;        // Riccati: the 12 N floats the gains overwrite, kept for a second sweep (riccati_keep_linear_terms)
  int32_t tile;        // byte tile starts here (double index)
  int32_t total_bytes;
  int32_t tile_w;      // row stride of the tile in bytes (power of two), 0: no tile
  int32_t tile_h;      // rows
  int32_t reach;       // R: the tile spans [m0 - R, m0 + R]
};

// The carve-up depends on control_steps and the L-BFGS memory only, so the control_steps
// specialisations of K1 evaluate it at compile time (offsets become immediates); the host uses the
// same function and adds the reach tile geometry.
// corners: the feasible set has corners (the vx/vy box cuts the max_vel_trans disc) -- the Riccati kernel then keeps room
// for a second sweep with another active set.
constexpr LdsLayout make_lds_layout(int n, int mem, bool riccati, bool corners = false) {
  LdsLayout l{};
  const int nv = 3 * n;
  int off = 0;
  l.prob = off; off += 32;
  l.state = off; off += 16;
  l.tol = off; off += 32;   // (= kTolDoubles, solver_context.h) stop tolerances + two cold per-instance constants (read once per iteration; kept out of the scalar
                            // registers), then the hop candidates of the current iteration (solver_context.h)
  l.term = off; off += 256;
  l.u = off; off += nv;
  l.gs = off; off += nv;
  l.gt = off; off += nv;
  l.gr = off; off += nv;
  // (Riccati: the direction takes the place of the total gradient -- the sweep has consumed it before the
  // forward pass writes d, and the next tangent-cone pass rewrites it after the candidates have read d)
  if (riccati) l.d = l.gt;
  else { l.d = off; off += nv; }
  if (riccati) {
    // the Riccati kernel keeps no previous iterate / gradient and no quasi-Newton pairs; the winner is
    // staged where the reduced gradient lived (disjoint lifetimes inside an iteration); of the per-step
    // float64 scratch it uses nx, ny and rt (disc curvature) only; its stage records and gains are float32
    l.u_new = l.gr;
    l.u_prev = l.u; l.gt_prev = l.u; l.S = off; l.Y = off; l.rho = off;
    l.cs = off; l.sn = off; l.dxs = off; l.dys = off; l.rx = off; l.ry = off;
    l.rt = off; off += n;
    l.nx = off; off += n;
    l.ny = off; off += n;
    l.mode = off; off += 2 * n;
    l.hess = off;
    off = (off + 1) & ~1;                   // (16-byte aligned: the records are read as 16-byte words)
    l.ric = off; off += 14 * n;             // 28 floats per stage, riccati.h
    l.keep = off; off += corners ? 6 * n : 0;   // (16-byte aligned: 14 n is even)
  } else {
    l.u_prev = off; off += nv;
    l.gt_prev = off; off += nv;
    l.u_new = off; off += nv;
    l.S = off; off += mem * nv;
    l.Y = off; off += mem * nv;
    l.rho = off; off += NEO_MPC_MAX_LBFGS_MEMORY;
    l.cs = off; off += n;
    l.sn = off; off += n;
    l.dxs = off; off += n;
    l.dys = off; off += n;
    l.rx = off; off += n;
    l.ry = off; off += n;
    l.rt = off; off += n;
    l.nx = off; off += n;
    l.ny = off; off += n;
    l.mode = off; off += 2 * n;  // int[4n]: mode, omega-frozen, near-kink, near-kink at the previous iterate
    l.hess = off; off += (nv <= 24) ? nv * nv : 0;
    l.ric = off;
    l.keep = off;
  }
  off = (off + 1) & ~1;        // 16-byte align the tile
  l.tile = off;
  l.total_bytes = off * 8;
  return l;
}

struct SolveArgs {
  const neo_mpc_problem* problems;
  neo_mpc_state* states;
  double* warm;
  neo_mpc_command* commands;
  double* solution;        // optional
  double* path;            // optional
  const double* footprints;  // optional
  const int32_t* success;    // postprocess only, optional
  double* velocities;        // optional packed [count][3] copy of the commands
  neo_mpc_state* states_out; // where K2 writes the state / the warm start back: the arrays they were read from
  double* warm_out;          // (device batches), or the caller's page-locked host arrays (neo_mpc_solve_batch)
  const double* term_table;  // [256] per-step costmap term by raw cell value
  const uint32_t* order;     // optional: workgroup w solves instance order[w] (neo_mpc_balance_dispatch_device); null: w
  uint32_t footprint_points;
  uint32_t count;
  DevParams p;
  DevMap map;
  LdsLayout lds;
};

struct ObjectiveArgs {
  const neo_mpc_problem* problems;
  const double* u;
  double* cost;
  const double* term_table;
  uint32_t count;
  DevParams p;
  DevMap map;
  double w_trans, w_orient, w_control, w_terminal, w_costmap;   // undivided, as the reference uses them (py:252-268)
};

struct IngestArgs {
  const uint8_t* src;  // raw nav2 costmap(s), row-major size_x * size_y each, back to back
  uint8_t* dst;        // padded map base (row 0 of the border) of the first map
  int32_t size_x, size_y, pitch, rows;  // rows = size_y + 2*border
  int32_t maps;        // number of maps (blockIdx.y)
  int32_t border;
  int64_t dst_stride;  // bytes between padded maps
};

struct CarrotArgs {
  neo_mpc_lookahead_params lp;
  neo_mpc_plan_batch b;  // device pointers
};

// A/B switches of the measurement tools: read from the environment ONCE, by neo_mpc_create (include/neo_mpc.h), kept in
// the handle -- nothing on the solve path looks at the environment.
struct LaunchTuning {
  int solve_waves = 0;         // NEO_MPC_SOLVE_WAVES=2|3|4: occupancy variant of K1 (0: what was measured fastest)
  bool no_tame = false;        // NEO_MPC_NO_TAME_SPECIALISATION: the general kernels for README-like parameters too
  bool dynamic_lds = false;    // NEO_MPC_DYNAMIC_LDS: the dynamic-LDS build of the control_steps specialisations
};

void launch_solve(const SolveArgs& a, const LaunchTuning& t, void* stream, void* ev_start = nullptr, void* ev_stop = nullptr);
void launch_carrots(const CarrotArgs& a, void* stream);
void launch_postprocess(const SolveArgs& a, void* stream);
void launch_objective(const ObjectiveArgs& a, void* stream);
void launch_ingest(const IngestArgs& a, const LaunchTuning& t, void* stream);
// K5: dispatch order of the next launch from the iteration counts of the previous one (neo_mpc_balance_dispatch_device)
void launch_dispatch_order(const neo_mpc_command* commands, float* load, uint32_t* order, uint32_t count, bool fresh, void* stream);
constexpr uint32_t kDispatchSimds = 1024;   // a 4096-instance launch is one residency round: workgroups w, w + 1024, w + 2048, w + 3072 share a SIMD

}  // namespace neo_mpc
