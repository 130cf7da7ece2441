// neo_mpc_capi.cpp -- the C-ABI of libneo_mpc.so (include/neo_mpc.h) over the gfx950 kernels.
//
// Host side of the drop-in boundary: what `NeoMpcPlanner::computeVelocityCommands`
// (src/NeoMpcPlanner.cpp:240-252) calls instead of the ROS2 service hop, and what
// `MpcOptimizationServer.__init__` (mpc_optimization_server.py:45-152) sets up.
// There is deliberately no CPU fallback: without a gfx950 device create() fails.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "neo_mpc_device.h"
#include "solver_rules.h"

using namespace neo_mpc;

namespace {

thread_local std::string g_error;
thread_local int g_error_code = 0;

int fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_error = buf;
  g_error_code = code;
  return code;
}

}  // namespace
// (shared with neo_mpc_rccl.cpp)
int neo_mpc_set_error(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_error = buf;
  g_error_code = code;
  return code;
}
namespace {

#define HIP_TRY(expr)                                                                   \
  do {                                                                                  \
    hipError_t e_ = (expr);                                                             \
    if (e_ != hipSuccess)                                                               \
      return fail(NEO_MPC_ERR_DEVICE, "%s failed: %s", #expr, hipGetErrorString(e_));   \
  } while (0)

struct DeviceBuffer {
  void* ptr = nullptr;
  size_t bytes = 0;
  int reserve(size_t need) {
    if (need <= bytes) return NEO_MPC_OK;
    if (ptr) (void)hipFree(ptr);
    ptr = nullptr;
    bytes = 0;
    size_t cap = need + need / 4 + 256;
    HIP_TRY(hipMalloc(&ptr, cap));
    bytes = cap;
    return NEO_MPC_OK;
  }
  void release() {
    if (ptr) (void)hipFree(ptr);
    ptr = nullptr;
    bytes = 0;
  }
};

}  // namespace

struct neo_mpc_handle {
  int device = 0;
  neo_mpc_params params{};
  DevParams dp{};
  DevMap map{};
  bool has_map = false;
  LdsLayout lds{};
  DeviceBuffer map_buf, raw_buf, term_buf, origins_buf;
  DeviceBuffer problems, states, warm, commands, solution, path, footprints, success, u, cost;
  DeviceBuffer plan_poses, plan_offsets, robot_poses, fp_costs, slow_down, carrots, vel;
  // latency path of neo_mpc_solve_batch (small host batches, the plugin's count = 1): one pinned
  // staging block and one device arena, so a tick is one H2D, K1, one D2H and one synchronisation
  void* pin = nullptr;
  DeviceBuffer arena;
  DeviceBuffer order_buf;     // dispatch order of device batches (neo_mpc_balance_dispatch_device)
  DeviceBuffer load_buf;      // ... and the exponential average of the iteration counts it is sorted by
  size_t order_count = 0;     // ... armed for batches of this many instances; 0: launch order
  // Stream ordering around the device map: every ingest records map_ready on the stream it ran on and
  // every solve / postprocess / objective launch waits for it on its own stream; every such launch
  // records the in-use event OF ITS STREAM (one per distinct stream the caller has used), and the next
  // ingest waits for all of them -- and for the previous ingest -- before it rewrites map_buf in place.
  hipEvent_t map_ready = nullptr;
  hipStream_t map_ready_stream = nullptr;   // (waits on the stream an event was recorded on are skipped: in order anyway)
  struct MapUser { hipStream_t stream; hipEvent_t done; bool pending; };
  std::vector<MapUser> map_users;
  int host_path = NEO_MPC_HOST_PATH_AUTO;   // neo_mpc_set_host_path
  // the environment's A/B switches, read once by neo_mpc_create (include/neo_mpc.h)
  LaunchTuning tuning;
  bool no_chunks = false;     // NEO_MPC_NO_CHUNKS: large staged host batches go through in one piece
  int auto_host_path = NEO_MPC_HOST_PATH_ZEROCOPY;   // what NEO_MPC_HOST_PATH_AUTO means (NEO_MPC_HOST_PATH)
  // neo_mpc_solve_batch_begin / _wait: page-locked batches in flight, each on a stream of its own
  struct InFlight { hipStream_t stream = nullptr; hipEvent_t done = nullptr; bool busy = false; };
  InFlight in_flight[NEO_MPC_MAX_BATCHES_IN_FLIGHT];
  hipStream_t chunk_streams[2] = {nullptr, nullptr};   // staged host batches of >= kChunkedMinCount instances: copy / solve pipeline
};
constexpr size_t kMaxMapUsers = 64;   // distinct streams with a launch in flight between two ingests
constexpr size_t kLatencyPathMaxCount = 64;
constexpr size_t kChunkedMinCount = 65536;   // staged host batches from here on go through in kChunks pieces on two streams
                                             // (measured: pageable 32 768 instances 24.0 M solves/s in pieces against 28.6 M in one,
                                             // 65 536: 38.9 / 31.6, 131 072: 42.6 / 32.5, 262 144: 45.3 / 33.2)
constexpr size_t kChunks = 4;
constexpr size_t kLatencyPathBytes = kLatencyPathMaxCount * (sizeof(neo_mpc_problem) + sizeof(neo_mpc_state) +
                                                            sizeof(neo_mpc_command) + 24 +
                                                            3 * 3 * NEO_MPC_MAX_CONTROL_STEPS * 8);

namespace {

int validate(const neo_mpc_params& p, bool live_handle) {
  if (p.control_steps < 1 || p.control_steps > NEO_MPC_MAX_CONTROL_STEPS)
    return fail(NEO_MPC_ERR_INVALID_ARGUMENT, "control_steps %d outside [1, %d]", p.control_steps,
                NEO_MPC_MAX_CONTROL_STEPS);
  if (!(p.prediction_horizon > 0.0)) return fail(NEO_MPC_ERR_INVALID_ARGUMENT, "prediction_horizon must be > 0");
  if (!(p.min_vel_x <= p.max_vel_x) || !(p.min_vel_y <= p.max_vel_y) || !(p.min_vel_theta <= p.max_vel_theta))
    return fail(NEO_MPC_ERR_INVALID_ARGUMENT, "velocity bounds: min > max");  // SciPy raises here too
  if (!(p.max_vel_trans > 0.0)) return fail(NEO_MPC_ERR_INVALID_ARGUMENT, "max_vel_trans must be > 0");
  // box ∩ disc must be non-empty: closest box point to the origin inside the disc
  double nx = std::fmin(std::fmax(0.0, p.min_vel_x), p.max_vel_x);
  double ny = std::fmin(std::fmax(0.0, p.min_vel_y), p.max_vel_y);
  if (nx * nx + ny * ny > p.max_vel_trans * p.max_vel_trans)
    return fail(NEO_MPC_ERR_INVALID_ARGUMENT, "velocity box does not intersect the max_vel_trans disc");
  if (p.method < 0 || p.method > NEO_MPC_METHOD_RICCATI) return fail(NEO_MPC_ERR_INVALID_ARGUMENT, "unknown method %d", p.method);
  if (p.method == NEO_MPC_METHOD_NEWTON && p.control_steps > NEO_MPC_NEWTON_MAX_CONTROL_STEPS)
    return fail(NEO_MPC_ERR_UNSUPPORTED, "NEO_MPC_METHOD_NEWTON is built for control_steps <= 8 only (got %d)",
                p.control_steps);
  if (p.lbfgs_memory > NEO_MPC_MAX_LBFGS_MEMORY)
    return fail(NEO_MPC_ERR_INVALID_ARGUMENT, "lbfgs_memory > %d", NEO_MPC_MAX_LBFGS_MEMORY);
  if (p.compat_flags & ~NEO_MPC_COMPAT_ALL)
    return fail(NEO_MPC_ERR_INVALID_ARGUMENT, "unknown compat_flags bits 0x%x (known: 0x%x)", p.compat_flags & ~NEO_MPC_COMPAT_ALL,
                NEO_MPC_COMPAT_ALL);
  // no selectable configuration is knowingly worse than the reference: the L-BFGS and the dense Newton direction have no
  // wall model for costmap steps, and forced onto a heavy costmap weight they end above SLSQP on a few percent of the
  // costmap cases (G8 "turn": up to 1.0 at control_steps 8; G9: 2 of 48 up to 4e-3) -- AUTO never sends them there
  // (neo_mpc_create refuses; a live handle reconfigured across the threshold runs the stage-wise direction instead --
  // cb_params cannot fail in the reference, and a controller must not lose its solver to a weight change: solver_rules.h)
  if (!live_handle && (p.method == NEO_MPC_METHOD_LBFGS || p.method == NEO_MPC_METHOD_NEWTON) && p.w_costmap > 0.25 * p.w_trans)
    return fail(NEO_MPC_ERR_UNSUPPORTED, "method %d has no wall model for costmap steps: not offered with w_costmap > w_trans / 4 "
                "(%g > %g); use NEO_MPC_METHOD_AUTO or NEO_MPC_METHOD_RICCATI", p.method, p.w_costmap, 0.25 * p.w_trans);
  return NEO_MPC_OK;
}

int occupancy(int raw) {  // nav2 Costmap2DPublisher translation (build's costmap contract)
  if (raw == 0) return 0;
  if (raw == 253) return 99;
  if (raw == 254) return 100;
  if (raw == 255) return -1;
  return 1 + (97 * (raw - 1)) / 251;
}

void derive(neo_mpc_handle* h) {
  const neo_mpc_params& p = h->params;
  DevParams& d = h->dp;
  const int n = p.control_steps;
  d.n = n;
  d.dt = p.prediction_horizon / n;  // py:137
  d.wt_n = p.w_trans / n;
  d.wo_n = p.w_orient / n;
  d.wc_n = p.w_control / n;
  d.wterm_o = p.w_terminal * p.w_orient;
  d.wterm_t = p.w_terminal * p.w_trans;
  d.w_footprint = p.w_footprint;
  d.lo[0] = p.min_vel_x; d.hi[0] = p.max_vel_x;
  d.lo[1] = p.min_vel_y; d.hi[1] = p.max_vel_y;
  d.lo[2] = p.min_vel_theta; d.hi[2] = p.max_vel_theta;
  d.r = p.max_vel_trans;
  d.acc[0] = p.acc_x_limit; d.acc[1] = p.acc_y_limit; d.acc[2] = p.acc_theta_limit;
  d.low_pass_gain = p.low_pass_gain;
  // every tolerance of the search comes out of solver_rules.h -- the one rule book the device code and the CPU mirror
  // (test infrastructure) share; all of them derive from opt_tolerance
  neo_rules r;
  neo_rules_derive(&p, &r);
  d.xtol = r.xtol;
  d.kink_radius = r.kink_radius;
  {
    neo_rules rs;   // (the rules of the instances the routed kernel sends to the stage-wise direction)
    neo_rules_derive_as(&p, NEO_DIRECTION_STAGEWISE, &rs);
    d.kink_radius_stagewise = rs.kink_radius;
  }
  d.stall_step = r.stall_step;
  d.hop_min_drop = r.hop_min_drop;
  d.hop_range = h->has_map ? neo_rules_hop_range(d.dt, h->map.resolution) : NEO_RULE_HOP_DIST;
  d.scan_resume_gain = r.scan_resume_gain;
  d.scan_reach = h->has_map ? neo_rules_reach_cells(&p, h->map.resolution) : 0;
  d.max_it = r.max_iterations;
  d.mem = r.lbfgs_memory;
  d.compat = (p.compat_flags & NEO_MPC_COMPAT_ODOM_YAW_GOAL_W) |
             ((p.compat_flags & NEO_MPC_COMPAT_REFERENCE_START) ? kCompatNoUnshift : 0);
  d.disc_in_box = (p.min_vel_x <= -p.max_vel_trans && p.max_vel_x >= p.max_vel_trans &&
                   p.min_vel_y <= -p.max_vel_trans && p.max_vel_y >= p.max_vel_trans) ? 1 : 0;
  d.tame = (d.disc_in_box && fmax(fabs(p.min_vel_theta), fabs(p.max_vel_theta)) * p.prediction_horizon <= 0.78) ? 1 : 0;
  // search direction: AUTO = the register-resident dense Newton kernel at control_steps 3 (the headline
  // specialisation), the stage-wise (Riccati) Newton sweep of the run-time-sized kernel otherwise -- and at 3 too
  // when the costmap weight is heavy (w_costmap > w_trans / 4; the README's is w_trans / 16): cost steps are then
  // walls the search has to slide along, and the wall model lives in the stage-wise direction (costmap.h).  Against
  // the reference's SLSQP solves at w_costmap = 0.3 (G8 "turn") the dense direction ends 3e-3 and 9e-3 above
  // SLSQP's value in 2 of 24 cases, the stage-wise one in none.
  d.newton = r.direction;
  // (round 6) AUTO at control_steps 3: direction by neighbourhood -- dense where the reach tile is all free, stage-wise
  // elsewhere (solver_rules.h neo_rules_routes_by_neighbourhood; method = NEO_MPC_METHOD_NEWTON is the dense direction for
  // every instance: round 5's AUTO, the A/B partner)
  d.routed = (neo_rules_routes_by_neighbourhood(&p) && r.direction == NEO_DIRECTION_DENSE) ? 1 : 0;
  d.early_tol = r.xtol;
  d.final_tol = r.final_tol;
  d.ftol = r.ftol;
  d.wtol = r.wtol;
  d.wtol_late = r.wtol_late;
  d.btol_map = r.btol_map;
  d.btol_free = r.btol_free;

  // LDS carve-up (shared with the kernel specialisations) + reach tile geometry
  LdsLayout& l = h->lds;
  // the control_steps == 3 specialisations carve LDS at compile time with 4 pair slots; the host
  // must reserve exactly that layout whenever launch_solve() will pick them
  const bool specialised = n == 3 && (d.newton == 1 || (d.newton == 0 && d.mem == 4));
  l = make_lds_layout(n, specialised ? 4 : (d.newton ? 0 : d.mem), d.newton == 2, !d.disc_in_box);
  const int off = l.tile;
  l.tile_w = 0; l.tile_h = 0; l.reach = 0;
  if (h->has_map) {
    const int R = neo_rules_reach_cells(&p, h->map.resolution);   // (solver_rules.h: the cell scan's radius too)
    const int w = neo_rules_tile_width(R);   // (0: no tile -- a reach beyond 60 cells)
    if (w) { l.reach = R; l.tile_w = w; l.tile_h = 2 * R + 1; }
  }
  l.total_bytes = off * 8 + l.tile_w * l.tile_h;
  l.total_bytes = (l.total_bytes + 15) & ~15;
}

// The term table and the pool's origins are read by every K1 wave as it starts: before either is rewritten (blocking
// copies on the null stream, which do not order against the non-blocking streams batches are in flight on) every launch
// that may still read them is waited for -- the events map_release recorded, one per stream.  (The device map itself is
// ordered by stream waits in ingest(); these two small tables change on reconfiguration / pool re-centring only, so a
// host-side wait is cheap.)
int wait_map_users(neo_mpc_handle* h) {
  for (auto& u : h->map_users)
    if (u.pending) HIP_TRY(hipEventSynchronize(u.done));
  return NEO_MPC_OK;
}

int upload_term_table(neo_mpc_handle* h) {
  double table[256];
  const neo_mpc_params& p = h->params;
  const int n = p.control_steps;
  for (int raw = 0; raw < 256; ++raw) {
    const double c = (double)occupancy(raw) / 100.0;  // getCost contract
    const double cc = c * c;                           // py:247
    table[raw] = (c == 1.0) ? cc * 1000 / n : p.w_costmap * cc / n;  // py:257-260
  }
  int rc = wait_map_users(h);   // (a batch in flight reads the old table to its end)
  if (rc) return rc;
  if ((rc = h->term_buf.reserve(sizeof(table)))) return rc;
  HIP_TRY(hipMemcpy(h->term_buf.ptr, table, sizeof(table), hipMemcpyHostToDevice));
  return NEO_MPC_OK;
}

int apply_params(neo_mpc_handle* h, const neo_mpc_params* params, bool live_handle) {
  int rc = validate(*params, live_handle);
  if (rc) return rc;
  h->params = *params;
  derive(h);
  return upload_term_table(h);
}

// `maps` raw costmaps back to back in device memory -> bordered, pitched device maps (K3).
// d_origins == nullptr: a single map with origin (ox, oy); else a pool whose origins stay where they are.
int ingest(neo_mpc_handle* h, const uint8_t* d_cells, uint32_t maps, uint32_t sx, uint32_t sy, double res, double ox,
           double oy, const double* d_origins, void* stream) {
  if (!d_cells || maps == 0 || sx == 0 || sy == 0 || sx > (1u << 20) || sy > (1u << 20) || !(res > 0.0))
    return fail(NEO_MPC_ERR_INVALID_ARGUMENT, "bad costmap geometry %ux%u x%u res %g", sx, sy, maps, res);
  const int border = d_origins ? kPoolBorder : kMapBorder;
  const int pitch = (int)((sx + 2 * border + 127) & ~127u);
  const int rows = (int)sy + 2 * border;
  const size_t stride = (size_t)pitch * rows;
  if (stride / 16 >= (1ull << 31))   // K3 indexes the 16-byte chunks of one map with 32 bits
    return fail(NEO_MPC_ERR_UNSUPPORTED, "costmap %ux%u is too large (over 32 GiB padded)", sx, sy);
  int rc = h->map_buf.reserve(stride * maps);
  if (rc) return rc;
  hipStream_t st = (hipStream_t)stream;
  if (!h->map_ready) HIP_TRY(hipEventCreateWithFlags(&h->map_ready, hipEventDisableTiming));
  // behind the previous ingest (two ingests on different streams must not overlap in map_buf) ...
  else if (h->map_ready_stream != st) HIP_TRY(hipStreamWaitEvent(st, h->map_ready, 0));
  // ... and behind every launch still reading the old map, whatever stream it went to
  for (auto& u : h->map_users)
    if (u.pending) {
      if (u.stream != st) HIP_TRY(hipStreamWaitEvent(st, u.done, 0));
      u.pending = false;
    }
  IngestArgs a;
  a.src = d_cells;
  a.dst = (uint8_t*)h->map_buf.ptr;
  a.size_x = (int)sx; a.size_y = (int)sy; a.pitch = pitch; a.rows = rows;
  a.maps = (int)maps; a.border = border; a.dst_stride = (int64_t)stride;
  launch_ingest(a, h->tuning, stream);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipEventRecord(h->map_ready, st));
  h->map_ready_stream = st;
  h->map.cells = (const uint8_t*)h->map_buf.ptr + (size_t)border * pitch + border;
  h->map.size_x = (int)sx; h->map.size_y = (int)sy; h->map.pitch = pitch;
  h->map.resolution = res; h->map.inv_resolution = 1.0 / res;
  h->map.origin_x = ox; h->map.origin_y = oy;
  h->map.pool_count = d_origins ? (int)maps : 0;
  h->map.border = border;
  h->map.pool_stride = (int64_t)stride;
  h->map.pool_origins = d_origins;
  h->has_map = true;
  derive(h);
  return NEO_MPC_OK;
}

int fill_args(neo_mpc_handle* h, const neo_mpc_batch* b, SolveArgs& a) {
  if (!h) return fail(NEO_MPC_ERR_INVALID_ARGUMENT, "null handle");
  if (!b) return fail(NEO_MPC_ERR_INVALID_ARGUMENT, "null batch");
  if (!h->has_map) return fail(NEO_MPC_ERR_NO_COSTMAP, "neo_mpc_set_costmap has not been called");
  if (b->count > 0 && (!b->problems || !b->states || !b->warm_start || !b->commands))
    return fail(NEO_MPC_ERR_INVALID_ARGUMENT, "problems/states/warm_start/commands must not be null");
  if (b->count > 0x7fffffffull) return fail(NEO_MPC_ERR_INVALID_ARGUMENT, "count too large");
  if (b->footprints && b->footprint_points > NEO_MPC_MAX_FOOTPRINT_POINTS)
    return fail(NEO_MPC_ERR_INVALID_ARGUMENT, "footprint_points > %d", NEO_MPC_MAX_FOOTPRINT_POINTS);
  std::memset(&a, 0, sizeof(a));
  a.problems = b->problems; a.states = b->states; a.warm = b->warm_start; a.commands = b->commands;
  a.solution = b->solution; a.path = b->predicted_path; a.velocities = b->velocities;
  a.states_out = b->states; a.warm_out = b->warm_start;
  a.footprints = b->footprint_points ? b->footprints : nullptr;
  a.footprint_points = b->footprints ? b->footprint_points : 0;
  a.count = (uint32_t)b->count;
  a.term_table = (const double*)h->term_buf.ptr;
  a.p = h->dp; a.map = h->map; a.lds = h->lds;
  return NEO_MPC_OK;
}

// order a launch that reads the device map on `stream`: behind the last ingest ...
int map_acquire(neo_mpc_handle* h, void* stream) {
  if (h->map_ready && h->map_ready_stream != (hipStream_t)stream)
    HIP_TRY(hipStreamWaitEvent((hipStream_t)stream, h->map_ready, 0));
  return NEO_MPC_OK;
}
// ... and in front of the next one: one event per distinct stream, re-recorded by that stream's latest launch
int map_release(neo_mpc_handle* h, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  neo_mpc_handle::MapUser* slot = nullptr;
  for (auto& u : h->map_users) if (u.stream == st) { slot = &u; break; }
  if (!slot) {
    if (h->map_users.size() >= kMaxMapUsers) {
      // more streams than slots: the oldest slot's launch is waited for here and the slot re-used
      slot = &h->map_users.front();
      if (slot->pending) HIP_TRY(hipEventSynchronize(slot->done));
      slot->stream = st;
    } else {
      hipEvent_t ev;
      HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
      h->map_users.push_back({st, ev, false});
      slot = &h->map_users.back();
    }
  }
  HIP_TRY(hipEventRecord(slot->done, st));
  slot->pending = true;
  return NEO_MPC_OK;
}

// host batch -> device staging; returns the device-pointer batch in `d`
int stage_in(neo_mpc_handle* h, const neo_mpc_batch* b, neo_mpc_batch& d, bool solution_is_input) {
  const size_t n = b->count, nv = 3 * (size_t)h->params.control_steps;
  int rc;
  if ((rc = h->problems.reserve(n * sizeof(neo_mpc_problem)))) return rc;
  if ((rc = h->states.reserve(n * sizeof(neo_mpc_state)))) return rc;
  if ((rc = h->warm.reserve(n * nv * 8))) return rc;
  if ((rc = h->commands.reserve(n * sizeof(neo_mpc_command)))) return rc;
  d = *b;
  d.problems = (const neo_mpc_problem*)h->problems.ptr;
  d.states = (neo_mpc_state*)h->states.ptr;
  d.warm_start = (double*)h->warm.ptr;
  d.commands = (neo_mpc_command*)h->commands.ptr;
  HIP_TRY(hipMemcpyAsync(h->problems.ptr, b->problems, n * sizeof(neo_mpc_problem), hipMemcpyHostToDevice, nullptr));
  HIP_TRY(hipMemcpyAsync(h->states.ptr, b->states, n * sizeof(neo_mpc_state), hipMemcpyHostToDevice, nullptr));
  HIP_TRY(hipMemcpyAsync(h->warm.ptr, b->warm_start, n * nv * 8, hipMemcpyHostToDevice, nullptr));
  if (b->solution) {
    if ((rc = h->solution.reserve(n * nv * 8))) return rc;
    d.solution = (double*)h->solution.ptr;
    if (solution_is_input) HIP_TRY(hipMemcpyAsync(h->solution.ptr, b->solution, n * nv * 8, hipMemcpyHostToDevice, nullptr));
  }
  if (b->predicted_path) {
    if ((rc = h->path.reserve(n * nv * 8))) return rc;
    d.predicted_path = (double*)h->path.ptr;
  }
  if (b->velocities) {
    if ((rc = h->vel.reserve(n * 24))) return rc;
    d.velocities = (double*)h->vel.ptr;
  }
  if (b->footprints && b->footprint_points) {
    const size_t bytes = n * b->footprint_points * 2 * 8;
    for (size_t k = 0; k < n * b->footprint_points * 2; ++k)
      if (!std::isfinite(b->footprints[k]))
        return fail(NEO_MPC_ERR_INVALID_ARGUMENT, "footprint vertex %zu of instance %zu is not finite",
                    (k / 2) % b->footprint_points, k / (2 * b->footprint_points));
    if ((rc = h->footprints.reserve(bytes))) return rc;
    HIP_TRY(hipMemcpyAsync(h->footprints.ptr, b->footprints, bytes, hipMemcpyHostToDevice, nullptr));
    d.footprints = (const double*)h->footprints.ptr;
  }
  return NEO_MPC_OK;
}

int stage_out(neo_mpc_handle* h, const neo_mpc_batch* b, bool solution_is_output) {
  const size_t n = b->count, nv = 3 * (size_t)h->params.control_steps;
  // every copy is queued on the null stream behind the kernel and waited for once: with page-locked host buffers
  // (hipHostMalloc / hipHostRegister / torch pin_memory) they are DMA transfers that overlap the host side of the
  // next call; with pageable buffers the runtime stages them and each call returns when its copy is done
  HIP_TRY(hipMemcpyAsync(b->commands, h->commands.ptr, n * sizeof(neo_mpc_command), hipMemcpyDeviceToHost, nullptr));
  HIP_TRY(hipMemcpyAsync(b->states, h->states.ptr, n * sizeof(neo_mpc_state), hipMemcpyDeviceToHost, nullptr));
  HIP_TRY(hipMemcpyAsync(b->warm_start, h->warm.ptr, n * nv * 8, hipMemcpyDeviceToHost, nullptr));
  if (b->solution && solution_is_output)
    HIP_TRY(hipMemcpyAsync(b->solution, h->solution.ptr, n * nv * 8, hipMemcpyDeviceToHost, nullptr));
  if (b->predicted_path)
    HIP_TRY(hipMemcpyAsync(b->predicted_path, h->path.ptr, n * nv * 8, hipMemcpyDeviceToHost, nullptr));
  if (b->velocities) HIP_TRY(hipMemcpyAsync(b->velocities, h->vel.ptr, n * 24, hipMemcpyDeviceToHost, nullptr));
  HIP_TRY(hipStreamSynchronize(nullptr));
  return NEO_MPC_OK;
}

}  // namespace

extern "C" {

int neo_mpc_abi_version(void) { return NEO_MPC_ABI_VERSION; }
int neo_mpc_behaviour_version(void) { return NEO_MPC_BEHAVIOUR_VERSION; }

const char* neo_mpc_last_error(void) { return g_error.c_str(); }
int neo_mpc_last_error_code(void) { return g_error_code; }

int neo_mpc_default_params(neo_mpc_params* p) {
  if (!p) return fail(NEO_MPC_ERR_INVALID_ARGUMENT, "null params");
  std::memset(p, 0, sizeof(*p));
  p->acc_x_limit = p->acc_y_limit = p->acc_theta_limit = 0.5;             // py:49-51
  p->min_vel_x = p->min_vel_y = p->min_vel_theta = -0.5;                   // py:53-56
  p->min_vel_trans = 0.5;                                                  // py:55 (sic)
  p->max_vel_x = p->max_vel_y = p->max_vel_trans = p->max_vel_theta = 0.5; // py:58-61
  p->w_trans = p->w_orient = p->w_control = p->w_terminal = p->w_costmap = 0.5;  // py:63-67
  p->w_footprint = 2000;                                                   // py:68
  p->waiting_time = 3.0;                                                   // py:70
  p->low_pass_gain = 0.5;                                                  // py:71
  p->opt_tolerance = 1e-5;                                                 // py:72
  p->prediction_horizon = 0.5;                                             // py:73
  p->control_steps = 3;                                                    // py:75
  p->max_iterations = 100;
  p->lbfgs_memory = 4;
  p->compat_flags = NEO_MPC_COMPAT_ODOM_YAW_GOAL_W;
  p->step_tolerance = 0.0;
  p->cost_tolerance = 0.0;
  p->kink_radius = 0.0;
  p->stall_step = 0.0;
  p->window_tolerance = 0.0;
  p->method = NEO_MPC_METHOD_AUTO;
  return NEO_MPC_OK;
}

neo_mpc_handle* neo_mpc_create(const neo_mpc_params* params, int device) {
  if (!params) { fail(NEO_MPC_ERR_INVALID_ARGUMENT, "null params"); return nullptr; }
  int count = 0;
  hipError_t e = hipGetDeviceCount(&count);
  if (e != hipSuccess || count <= 0) {
    fail(NEO_MPC_ERR_NO_DEVICE, "no HIP device visible (%s); this library has no CPU fallback",
         e == hipSuccess ? "device count 0" : hipGetErrorString(e));
    return nullptr;
  }
  if (device < 0 || device >= count) { fail(NEO_MPC_ERR_INVALID_ARGUMENT, "device %d of %d", device, count); return nullptr; }
  if (hipSetDevice(device) != hipSuccess) { fail(NEO_MPC_ERR_DEVICE, "hipSetDevice(%d) failed", device); return nullptr; }
  neo_mpc_handle* h = new (std::nothrow) neo_mpc_handle();
  if (!h) { fail(NEO_MPC_ERR_DEVICE, "out of host memory"); return nullptr; }
  h->device = device;
  {  // the A/B switches of the measurement tools: the only place the library looks at the environment
    const char* e = getenv("NEO_MPC_SOLVE_WAVES");
    h->tuning.solve_waves = (e && atoi(e) >= 2 && atoi(e) <= 4) ? atoi(e) : 0;
    h->tuning.no_tame = getenv("NEO_MPC_NO_TAME_SPECIALISATION") != nullptr;
    h->tuning.dynamic_lds = getenv("NEO_MPC_DYNAMIC_LDS") != nullptr;
    h->no_chunks = getenv("NEO_MPC_NO_CHUNKS") != nullptr;
    e = getenv("NEO_MPC_HOST_PATH");
    h->auto_host_path = !e ? NEO_MPC_HOST_PATH_ZEROCOPY : !strcmp(e, "staged") ? NEO_MPC_HOST_PATH_STAGED
                        : !strcmp(e, "zerocopy_out") ? NEO_MPC_HOST_PATH_ZEROCOPY_OUT : NEO_MPC_HOST_PATH_ZEROCOPY;
  }
  if (apply_params(h, params, false) != NEO_MPC_OK) { neo_mpc_destroy(h); return nullptr; }
  return h;
}

void neo_mpc_destroy(neo_mpc_handle* h) {
  if (!h) return;
  (void)hipSetDevice(h->device);
  DeviceBuffer* all[] = {&h->map_buf, &h->raw_buf, &h->term_buf, &h->problems, &h->states, &h->warm, &h->commands,
                         &h->solution, &h->path, &h->footprints, &h->success, &h->u, &h->cost, &h->plan_poses,
                         &h->plan_offsets, &h->robot_poses, &h->fp_costs, &h->slow_down, &h->carrots, &h->vel,
                         &h->arena, &h->origins_buf, &h->order_buf, &h->load_buf};
  for (DeviceBuffer* b : all) b->release();
  if (h->pin) (void)hipHostFree(h->pin);
  if (h->map_ready) (void)hipEventDestroy(h->map_ready);
  for (auto& u : h->map_users) (void)hipEventDestroy(u.done);
  for (hipStream_t cs : h->chunk_streams) if (cs) (void)hipStreamDestroy(cs);
  for (auto& f : h->in_flight) {
    if (f.busy) (void)hipEventSynchronize(f.done);
    if (f.done) (void)hipEventDestroy(f.done);
    if (f.stream) (void)hipStreamDestroy(f.stream);
  }
  delete h;
}

int neo_mpc_set_params(neo_mpc_handle* h, const neo_mpc_params* params) {
  if (!h || !params) return fail(NEO_MPC_ERR_INVALID_ARGUMENT, "null argument");
  HIP_TRY(hipSetDevice(h->device));
  return apply_params(h, params, true);
}

int neo_mpc_effective_method(const neo_mpc_handle* h) {
  if (!h) return fail(NEO_MPC_ERR_INVALID_ARGUMENT, "null handle");
  const int d = neo_rules_direction(&h->params);
  return d == NEO_DIRECTION_LBFGS ? NEO_MPC_METHOD_LBFGS : d == NEO_DIRECTION_DENSE ? NEO_MPC_METHOD_NEWTON : NEO_MPC_METHOD_RICCATI;
}

int neo_mpc_get_params(const neo_mpc_handle* h, neo_mpc_params* params) {
  if (!h || !params) return fail(NEO_MPC_ERR_INVALID_ARGUMENT, "null argument");
  *params = h->params;
  return NEO_MPC_OK;
}

int neo_mpc_set_costmap(neo_mpc_handle* h, const uint8_t* cells, uint32_t sx, uint32_t sy, double res, double ox,
                        double oy) {
  if (!h || !cells) return fail(NEO_MPC_ERR_INVALID_ARGUMENT, "null argument");
  HIP_TRY(hipSetDevice(h->device));
  int rc = h->raw_buf.reserve((size_t)sx * sy);
  if (rc) return rc;
  HIP_TRY(hipMemcpy(h->raw_buf.ptr, cells, (size_t)sx * sy, hipMemcpyHostToDevice));
  // no synchronisation: `cells` has been consumed by the (blocking) copy above; K3 runs on the null
  // stream and every later launch waits for map_ready (recorded by ingest) on its own stream
  return ingest(h, (const uint8_t*)h->raw_buf.ptr, 1, sx, sy, res, ox, oy, nullptr, nullptr);
}

int neo_mpc_set_costmap_device(neo_mpc_handle* h, const uint8_t* d_cells, uint32_t sx, uint32_t sy, double res,
                               double ox, double oy, void* stream) {
  if (!h || !d_cells) return fail(NEO_MPC_ERR_INVALID_ARGUMENT, "null argument");
  HIP_TRY(hipSetDevice(h->device));
  return ingest(h, d_cells, 1, sx, sy, res, ox, oy, nullptr, stream);
}

int neo_mpc_set_costmap_pool_device(neo_mpc_handle* h, const uint8_t* d_cells, uint32_t count, uint32_t sx,
                                    uint32_t sy, double res, const double* d_origins, void* stream) {
  if (!h || !d_cells || !d_origins) return fail(NEO_MPC_ERR_INVALID_ARGUMENT, "null argument");
  if (count == 0) return fail(NEO_MPC_ERR_INVALID_ARGUMENT, "bad costmap count %u", count);
  if (count > NEO_MPC_MAX_POOL_MAPS)   // K3 takes the map index from the grid's y coordinate
    return fail(NEO_MPC_ERR_UNSUPPORTED, "costmap pool of %u maps: at most %u per call", count, NEO_MPC_MAX_POOL_MAPS);
  HIP_TRY(hipSetDevice(h->device));
  return ingest(h, d_cells, count, sx, sy, res, 0.0, 0.0, d_origins, stream);
}

int neo_mpc_set_costmap_pool(neo_mpc_handle* h, const uint8_t* cells, uint32_t count, uint32_t sx, uint32_t sy,
                             double res, const double* origins) {
  if (!h || !cells || !origins) return fail(NEO_MPC_ERR_INVALID_ARGUMENT, "null argument");
  if (count == 0) return fail(NEO_MPC_ERR_INVALID_ARGUMENT, "bad costmap count %u", count);
  if (count > NEO_MPC_MAX_POOL_MAPS)   // K3 takes the map index from the grid's y coordinate
    return fail(NEO_MPC_ERR_UNSUPPORTED, "costmap pool of %u maps: at most %u per call", count, NEO_MPC_MAX_POOL_MAPS);
  HIP_TRY(hipSetDevice(h->device));
  const size_t bytes = (size_t)sx * sy * count;
  int rc = h->raw_buf.reserve(bytes);
  if (rc) return rc;
  if ((rc = wait_map_users(h))) return rc;   // (batches in flight pair the old origins with the old cells to their end)
  if ((rc = h->origins_buf.reserve((size_t)count * 16))) return rc;
  HIP_TRY(hipMemcpy(h->raw_buf.ptr, cells, bytes, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(h->origins_buf.ptr, origins, (size_t)count * 16, hipMemcpyHostToDevice));
  return ingest(h, (const uint8_t*)h->raw_buf.ptr, count, sx, sy, res, 0.0, 0.0, (const double*)h->origins_buf.ptr,
                nullptr);
}

int neo_mpc_solve_batch_device_timed(neo_mpc_handle* h, const neo_mpc_batch* batch, void* stream, void* start_event,
                                     void* stop_event) {
  SolveArgs a;
  int rc = fill_args(h, batch, a);
  if (rc) return rc;
  HIP_TRY(hipSetDevice(h->device));  // the stream and the buffers must belong to the handle's device
  if ((rc = map_acquire(h, stream))) return rc;
  if (h->order_count != 0 && h->order_count == batch->count) a.order = (const uint32_t*)h->order_buf.ptr;
  launch_solve(a, h->tuning, stream, start_event, stop_event);
  HIP_TRY(hipGetLastError());
  return map_release(h, stream);
}

// Balanced dispatch for a fleet's next tick (K5): see k_dispatch_order.  Results do not depend on it.
int neo_mpc_balance_dispatch_device(neo_mpc_handle* h, const neo_mpc_command* d_previous_commands, size_t count, void* stream) {
  if (!h) return fail(NEO_MPC_ERR_INVALID_ARGUMENT, "null handle");
  if (!d_previous_commands || count == 0) { h->order_count = 0; return NEO_MPC_OK; }
  if (count > 0xffffffffull) return fail(NEO_MPC_ERR_INVALID_ARGUMENT, "count too large");
  HIP_TRY(hipSetDevice(h->device));
  // (the average is kept while calls of the same count follow one another; disarming or another count starts it afresh)
  const bool fresh = h->order_count != count;
  // disarmed until the new order is on its way: a reserve() that re-allocates for a larger count and then fails, or a launch
  // that fails, must not leave an order armed that points at a buffer nobody has written
  h->order_count = 0;
  int rc = h->order_buf.reserve(count * sizeof(uint32_t));
  if (rc) return rc;
  if ((rc = h->load_buf.reserve(count * sizeof(float)))) return rc;
  launch_dispatch_order(d_previous_commands, (float*)h->load_buf.ptr, (uint32_t*)h->order_buf.ptr, (uint32_t)count, fresh, stream);
  HIP_TRY(hipGetLastError());
  h->order_count = count;
  return NEO_MPC_OK;
}

int neo_mpc_solve_batch_device(neo_mpc_handle* h, const neo_mpc_batch* batch, void* stream) {
  return neo_mpc_solve_batch_device_timed(h, batch, stream, nullptr, nullptr);
}

// neo_mpc_solve_batch for count <= kLatencyPathMaxCount without a footprint raster:
// [problems | states | warm | commands | velocities | solution | path] laid out once in a pinned block and
// mirrored in a device arena -- the whole input goes up in one copy, the whole output comes back in
// one, and the only synchronisation is the one that makes the results visible.
static int solve_batch_latency_path(neo_mpc_handle* h, const neo_mpc_batch* b) {
  const size_t n = b->count, nv = 3 * (size_t)h->params.control_steps;
  if (!h->pin) HIP_TRY(hipHostMalloc(&h->pin, kLatencyPathBytes, hipHostMallocDefault));
  int rc = h->arena.reserve(kLatencyPathBytes);
  if (rc) return rc;
  const size_t o_prob = 0, o_state = o_prob + n * sizeof(neo_mpc_problem), o_warm = o_state + n * sizeof(neo_mpc_state),
               o_cmd = o_warm + n * nv * 8, o_vel = o_cmd + n * sizeof(neo_mpc_command), o_sol = o_vel + n * 24,
               o_path = o_sol + (b->solution ? n * nv * 8 : 0), o_end = o_path + (b->predicted_path ? n * nv * 8 : 0);
  char* pin = (char*)h->pin;
  char* dev = (char*)h->arena.ptr;
  memcpy(pin + o_prob, b->problems, n * sizeof(neo_mpc_problem));
  memcpy(pin + o_state, b->states, n * sizeof(neo_mpc_state));
  memcpy(pin + o_warm, b->warm_start, n * nv * 8);
  HIP_TRY(hipMemcpyAsync(dev, pin, o_cmd, hipMemcpyHostToDevice, nullptr));
  neo_mpc_batch d = *b;
  d.problems = (const neo_mpc_problem*)(dev + o_prob);
  d.states = (neo_mpc_state*)(dev + o_state);
  d.warm_start = (double*)(dev + o_warm);
  d.commands = (neo_mpc_command*)(dev + o_cmd);
  d.velocities = (double*)(dev + o_vel);
  if (b->solution) d.solution = (double*)(dev + o_sol);
  if (b->predicted_path) d.predicted_path = (double*)(dev + o_path);
  SolveArgs a;
  if ((rc = fill_args(h, &d, a))) return rc;
  if ((rc = map_acquire(h, nullptr))) return rc;
  launch_solve(a, h->tuning, nullptr);
  HIP_TRY(hipGetLastError());
  if ((rc = map_release(h, nullptr))) return rc;
  HIP_TRY(hipMemcpyAsync(pin + o_state, dev + o_state, o_end - o_state, hipMemcpyDeviceToHost, nullptr));
  HIP_TRY(hipStreamSynchronize(nullptr));
  memcpy(b->states, pin + o_state, n * sizeof(neo_mpc_state));
  memcpy(b->warm_start, pin + o_warm, n * nv * 8);
  memcpy(b->commands, pin + o_cmd, n * sizeof(neo_mpc_command));
  if (b->velocities) memcpy(b->velocities, pin + o_vel, n * 24);
  if (b->solution) memcpy(b->solution, pin + o_sol, n * nv * 8);
  if (b->predicted_path) memcpy(b->predicted_path, pin + o_path, n * nv * 8);
  return NEO_MPC_OK;
}

// Is `p` page-locked host memory the device can address (hipHostMalloc / hipHostRegister / neo_mpc_pin_host_memory /
// torch pin_memory)?  -> its device-side address.
static bool pinned_host(const void* p, size_t bytes, void** dev) {
  hipPointerAttribute_t at;
  if (hipPointerGetAttributes(&at, p) != hipSuccess) { (void)hipGetLastError(); return false; }   // (unregistered memory: an error before ROCm 6)
  if (at.type != hipMemoryTypeHost || !at.devicePointer) return false;
  *dev = at.devicePointer;
  if (bytes > 1) {
    // ... to its LAST byte (a registered prefix of an arena, or count * stride beyond the pinned range, would fault on the
    // GPU instead of falling back to staging), and as ONE mapping: the device-side addresses are as far apart as the host's
    hipPointerAttribute_t last;
    const char* end = static_cast<const char*>(p) + bytes - 1;
    if (hipPointerGetAttributes(&last, end) != hipSuccess) { (void)hipGetLastError(); return false; }
    if (last.type != hipMemoryTypeHost || !last.devicePointer) return false;
    if (static_cast<const char*>(last.devicePointer) - static_cast<const char*>(at.devicePointer) != (ptrdiff_t)(bytes - 1)) return false;
  }
  return true;
}

// neo_mpc_solve_batch when every array of the batch is page-locked: no staging copy at all.
//  kZeroCopy  K1 reads the request / state / warm-start records straight from the caller's arrays over PCIe -- they
//             are read once, at the start of each wave, as coalesced 8-byte-per-lane loads -- and writes commands,
//             state, warm start (and solution / path / velocities) straight back; the link is busy while other
//             waves compute.  One launch, one wait.
//  kZeroCopyOut  the inputs go up as three DMA copies into device staging, the results are written straight into the
//             caller's arrays by K1 (no D2H copies).
// NEO_MPC_HOST_PATH=staged|zerocopy|zerocopy_out in the environment of neo_mpc_create overrides what AUTO means (A/B runs).
enum HostPath { kStaged = NEO_MPC_HOST_PATH_STAGED, kZeroCopy = NEO_MPC_HOST_PATH_ZEROCOPY,
                kZeroCopyOut = NEO_MPC_HOST_PATH_ZEROCOPY_OUT };
static HostPath host_path_mode(const neo_mpc_handle* h) {
  return (HostPath)(h->host_path != NEO_MPC_HOST_PATH_AUTO ? h->host_path : h->auto_host_path);
}

static int solve_batch_zero_copy(neo_mpc_handle* h, const neo_mpc_batch* b, const neo_mpc_batch& dev, HostPath mode) {
  const size_t n = b->count, nv = 3 * (size_t)h->params.control_steps;
  neo_mpc_batch d = dev;          // every pointer: the device-side address of the caller's page-locked array
  SolveArgs a;
  int rc;
  auto bail = [](int code) { (void)hipStreamSynchronize(nullptr); return code; };
  if (mode == kZeroCopyOut) {
    if ((rc = h->problems.reserve(n * sizeof(neo_mpc_problem)))) return rc;
    if ((rc = h->states.reserve(n * sizeof(neo_mpc_state)))) return rc;
    if ((rc = h->warm.reserve(n * nv * 8))) return rc;
    HIP_TRY(hipMemcpyAsync(h->problems.ptr, b->problems, n * sizeof(neo_mpc_problem), hipMemcpyHostToDevice, nullptr));
    HIP_TRY(hipMemcpyAsync(h->states.ptr, b->states, n * sizeof(neo_mpc_state), hipMemcpyHostToDevice, nullptr));
    HIP_TRY(hipMemcpyAsync(h->warm.ptr, b->warm_start, n * nv * 8, hipMemcpyHostToDevice, nullptr));
    d.problems = (const neo_mpc_problem*)h->problems.ptr;
    d.states = (neo_mpc_state*)h->states.ptr;
    d.warm_start = (double*)h->warm.ptr;
  }
  if ((rc = fill_args(h, &d, a))) return bail(rc);
  a.states_out = dev.states; a.warm_out = dev.warm_start;
  if ((rc = map_acquire(h, nullptr))) return bail(rc);
  launch_solve(a, h->tuning, nullptr);
  if (hipGetLastError() != hipSuccess) return bail(fail(NEO_MPC_ERR_DEVICE, "kernel launch failed"));
  if ((rc = map_release(h, nullptr))) return bail(rc);
  HIP_TRY(hipStreamSynchronize(nullptr));   // kernel end = system-scope release: the results are in the caller's arrays
  return NEO_MPC_OK;
}

// Is every array of the host batch page-locked over its whole extent?  -> `dv`: the batch with the device-side addresses.
static bool batch_page_locked(const neo_mpc_handle* h, const neo_mpc_batch* batch, neo_mpc_batch& dv) {
  dv = *batch;
  const size_t n = batch->count, nv = 3 * (size_t)h->params.control_steps;
  bool all = pinned_host(batch->problems, n * sizeof(neo_mpc_problem), (void**)&dv.problems) &&
             pinned_host(batch->states, n * sizeof(neo_mpc_state), (void**)&dv.states) &&
             pinned_host(batch->warm_start, n * nv * 8, (void**)&dv.warm_start) &&
             pinned_host(batch->commands, n * sizeof(neo_mpc_command), (void**)&dv.commands);
  if (all && batch->solution) all = pinned_host(batch->solution, n * nv * 8, (void**)&dv.solution);
  if (all && batch->predicted_path) all = pinned_host(batch->predicted_path, n * nv * 8, (void**)&dv.predicted_path);
  if (all && batch->velocities) all = pinned_host(batch->velocities, n * 24, (void**)&dv.velocities);
  if (all && batch->footprints && batch->footprint_points)
    all = pinned_host(batch->footprints, n * batch->footprint_points * 16, (void**)&dv.footprints);
  return all;
}

// A staged host batch of kChunkedMinCount instances or more goes through in kChunks pieces on two streams: piece c + 1
// is copied up and launched before piece c's results are copied back, so copies and kernels of neighbouring pieces
// overlap (with pageable arrays the runtime's bounce copies block the host, the kernels run behind them; with
// page-locked arrays everything is asynchronous).  The instances are independent: the pieces' results are the batch's.
static int solve_batch_staged_chunks(neo_mpc_handle* h, const neo_mpc_batch* b) {
  const size_t n = b->count, nv = 3 * (size_t)h->params.control_steps;
  int rc;
  if ((rc = h->problems.reserve(n * sizeof(neo_mpc_problem)))) return rc;
  if ((rc = h->states.reserve(n * sizeof(neo_mpc_state)))) return rc;
  if ((rc = h->warm.reserve(n * nv * 8))) return rc;
  if ((rc = h->commands.reserve(n * sizeof(neo_mpc_command)))) return rc;
  if (b->solution && (rc = h->solution.reserve(n * nv * 8))) return rc;
  if (b->predicted_path && (rc = h->path.reserve(n * nv * 8))) return rc;
  if (b->velocities && (rc = h->vel.reserve(n * 24))) return rc;
  for (hipStream_t& cs : h->chunk_streams)
    if (!cs) HIP_TRY(hipStreamCreateWithFlags(&cs, hipStreamNonBlocking));
  auto bail = [&](int code) {
    for (hipStream_t cs : h->chunk_streams) (void)hipStreamSynchronize(cs);
    return code;
  };
  const size_t per = ((n + kChunks - 1) / kChunks + 63) & ~(size_t)63;
  auto piece = [&](size_t c, size_t& off, size_t& m) { off = c * per; m = off < n ? (n - off < per ? n - off : per) : 0; };
  auto up_and_launch = [&](size_t c) -> int {
    size_t off, m;
    piece(c, off, m);
    if (!m) return NEO_MPC_OK;
    hipStream_t st = h->chunk_streams[c & 1];
    char* d_prob = (char*)h->problems.ptr + off * sizeof(neo_mpc_problem);
    char* d_state = (char*)h->states.ptr + off * sizeof(neo_mpc_state);
    double* d_warm = (double*)h->warm.ptr + off * nv;
    HIP_TRY(hipMemcpyAsync(d_prob, b->problems + off, m * sizeof(neo_mpc_problem), hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(d_state, b->states + off, m * sizeof(neo_mpc_state), hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(d_warm, b->warm_start + off * nv, m * nv * 8, hipMemcpyHostToDevice, st));
    neo_mpc_batch d = *b;
    d.count = m;
    d.problems = (const neo_mpc_problem*)d_prob;
    d.states = (neo_mpc_state*)d_state;
    d.warm_start = d_warm;
    d.commands = (neo_mpc_command*)h->commands.ptr + off;
    d.solution = b->solution ? (double*)h->solution.ptr + off * nv : nullptr;
    d.predicted_path = b->predicted_path ? (double*)h->path.ptr + off * nv : nullptr;
    d.velocities = b->velocities ? (double*)h->vel.ptr + off * 3 : nullptr;
    d.footprints = nullptr; d.footprint_points = 0;
    SolveArgs a;
    int r = fill_args(h, &d, a);
    if (r) return r;
    if ((r = map_acquire(h, st))) return r;
    launch_solve(a, h->tuning, st);
    if (hipGetLastError() != hipSuccess) return fail(NEO_MPC_ERR_DEVICE, "kernel launch failed");
    return map_release(h, st);
  };
  auto down = [&](size_t c) -> int {
    size_t off, m;
    piece(c, off, m);
    if (!m) return NEO_MPC_OK;
    hipStream_t st = h->chunk_streams[c & 1];
    HIP_TRY(hipMemcpyAsync(b->commands + off, (neo_mpc_command*)h->commands.ptr + off, m * sizeof(neo_mpc_command), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(b->states + off, (neo_mpc_state*)h->states.ptr + off, m * sizeof(neo_mpc_state), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(b->warm_start + off * nv, (double*)h->warm.ptr + off * nv, m * nv * 8, hipMemcpyDeviceToHost, st));
    if (b->solution) HIP_TRY(hipMemcpyAsync(b->solution + off * nv, (double*)h->solution.ptr + off * nv, m * nv * 8, hipMemcpyDeviceToHost, st));
    if (b->predicted_path)
      HIP_TRY(hipMemcpyAsync(b->predicted_path + off * nv, (double*)h->path.ptr + off * nv, m * nv * 8, hipMemcpyDeviceToHost, st));
    if (b->velocities) HIP_TRY(hipMemcpyAsync(b->velocities + off * 3, (double*)h->vel.ptr + off * 3, m * 24, hipMemcpyDeviceToHost, st));
    return NEO_MPC_OK;
  };
  if ((rc = up_and_launch(0))) return bail(rc);
  for (size_t c = 0; c < kChunks; ++c) {
    if (c + 1 < kChunks && (rc = up_and_launch(c + 1))) return bail(rc);
    if ((rc = down(c))) return bail(rc);
  }
  for (hipStream_t cs : h->chunk_streams) HIP_TRY(hipStreamSynchronize(cs));
  return NEO_MPC_OK;
}

// neo_mpc_problem.skip was reserved[0] in ABI 1, whose header never asked for zeroed reserved bytes: a host batch whose
// skip fields hold anything but 0 or 1 is refused instead of having robots silently left out (one int per record; device
// batches cannot be scanned -- K1 and K2 act on skip == 1 alone there).
static int check_skip_fields(const neo_mpc_batch* batch) {
  for (size_t i = 0; i < batch->count; ++i)
    if ((uint32_t)batch->problems[i].skip > 1u)
      return fail(NEO_MPC_ERR_INVALID_ARGUMENT, "problems[%zu].skip = %d: must be 0 or 1 (reserved fields must be zeroed)", i,
                  batch->problems[i].skip);
  return NEO_MPC_OK;
}

int neo_mpc_solve_batch(neo_mpc_handle* h, const neo_mpc_batch* batch) {
  SolveArgs a;
  int rc = fill_args(h, batch, a);  // validates
  if (rc) return rc;
  if (batch->count == 0) return NEO_MPC_OK;
  if ((rc = check_skip_fields(batch))) return rc;
  HIP_TRY(hipSetDevice(h->device));
  if (batch->count <= kLatencyPathMaxCount && !batch->footprints)
    return solve_batch_latency_path(h, batch);
  if (host_path_mode(h) != kStaged) {
    // page-locked arrays throughout (a fleet server's request arena): K1 works on them in place
    neo_mpc_batch dv;
    if (batch_page_locked(h, batch, dv)) return solve_batch_zero_copy(h, batch, dv, host_path_mode(h));
  }
  if (batch->count >= kChunkedMinCount && !(batch->footprints && batch->footprint_points) && !h->no_chunks)
    return solve_batch_staged_chunks(h, batch);
  neo_mpc_batch d;
  // (the staging copies are asynchronous: no way out of here while one may still be reading the caller's buffers)
  auto bail = [](int code) { (void)hipStreamSynchronize(nullptr); return code; };
  if ((rc = stage_in(h, batch, d, false))) return bail(rc);
  if ((rc = fill_args(h, &d, a))) return bail(rc);
  if ((rc = map_acquire(h, nullptr))) return bail(rc);
  launch_solve(a, h->tuning, nullptr);
  if (hipGetLastError() != hipSuccess) return bail(fail(NEO_MPC_ERR_DEVICE, "kernel launch failed"));
  if ((rc = map_release(h, nullptr))) return bail(rc);
  return bail(stage_out(h, batch, true));   // (queued behind the kernel on the null stream, one wait at the end)
}

// The two halves of the reference's `async_send_request(request)` ... `result.get()` (cpp:248-250) for a batch: begin
// enqueues K1 on a stream of its own -- the batch's arrays are page-locked and worked on in place, as in
// neo_mpc_solve_batch -- and returns; wait blocks until that batch's results are in the caller's arrays.  With two
// batches in flight the launch latency, the wait for the records over PCIe and the host's own work between calls
// hide under the other batch's kernel.
int neo_mpc_solve_batch_begin(neo_mpc_handle* h, const neo_mpc_batch* batch, uint32_t* ticket) {
  if (!ticket) return fail(NEO_MPC_ERR_INVALID_ARGUMENT, "null ticket");
  *ticket = 0;
  SolveArgs a;
  int rc = fill_args(h, batch, a);  // validates
  if (rc) return rc;
  if (batch->count == 0) return NEO_MPC_OK;   // (ticket 0: nothing to wait for)
  if ((rc = check_skip_fields(batch))) return rc;
  HIP_TRY(hipSetDevice(h->device));
  neo_mpc_batch dv;
  if (!batch_page_locked(h, batch, dv))
    return fail(NEO_MPC_ERR_UNSUPPORTED, "neo_mpc_solve_batch_begin works on page-locked arrays in place: pin the batch's "
                "arrays (neo_mpc_pin_host_memory) or call neo_mpc_solve_batch");
  neo_mpc_handle::InFlight* slot = nullptr;
  uint32_t index = 0;
  for (uint32_t k = 0; k < NEO_MPC_MAX_BATCHES_IN_FLIGHT; ++k)
    if (!h->in_flight[k].busy) { slot = &h->in_flight[k]; index = k; break; }
  if (!slot) return fail(NEO_MPC_ERR_INVALID_ARGUMENT, "%d batches in flight already: wait for one", NEO_MPC_MAX_BATCHES_IN_FLIGHT);
  if (!slot->stream) HIP_TRY(hipStreamCreateWithFlags(&slot->stream, hipStreamNonBlocking));
  // (system-scope release: the results are visible to the host after hipEventSynchronize also in non-coherent pinned
  // memory -- hipHostMallocNonCoherent, HIP_HOST_COHERENT=0)
  if (!slot->done) HIP_TRY(hipEventCreateWithFlags(&slot->done, hipEventDisableTiming | hipEventReleaseToSystem));
  if ((rc = fill_args(h, &dv, a))) return rc;
  a.states_out = dv.states; a.warm_out = dv.warm_start;
  if ((rc = map_acquire(h, slot->stream))) return rc;
  launch_solve(a, h->tuning, slot->stream);
  // from here on a kernel may be writing the caller's arrays: no way out without a ticket unless it has been waited for
  auto bail = [&](int code) { (void)hipStreamSynchronize(slot->stream); return code; };
  if (hipGetLastError() != hipSuccess) return bail(fail(NEO_MPC_ERR_DEVICE, "kernel launch failed"));
  if ((rc = map_release(h, slot->stream))) return bail(rc);
  if (hipEventRecord(slot->done, slot->stream) != hipSuccess) return bail(fail(NEO_MPC_ERR_DEVICE, "hipEventRecord failed"));
  slot->busy = true;
  *ticket = index + 1;
  return NEO_MPC_OK;
}

int neo_mpc_solve_batch_wait(neo_mpc_handle* h, uint32_t ticket) {
  if (!h) return fail(NEO_MPC_ERR_INVALID_ARGUMENT, "null handle");
  if (ticket == 0) return NEO_MPC_OK;
  if (ticket > NEO_MPC_MAX_BATCHES_IN_FLIGHT || !h->in_flight[ticket - 1].busy)
    return fail(NEO_MPC_ERR_INVALID_ARGUMENT, "ticket %u names no batch in flight", ticket);
  neo_mpc_handle::InFlight& f = h->in_flight[ticket - 1];
  f.busy = false;
  HIP_TRY(hipEventSynchronize(f.done));   // kernel end = system-scope release: the results are in the caller's arrays
  return NEO_MPC_OK;
}

int neo_mpc_set_host_path(neo_mpc_handle* h, int mode) {
  if (!h) return fail(NEO_MPC_ERR_INVALID_ARGUMENT, "null handle");
  if (mode < NEO_MPC_HOST_PATH_AUTO || mode > NEO_MPC_HOST_PATH_ZEROCOPY_OUT)
    return fail(NEO_MPC_ERR_INVALID_ARGUMENT, "unknown host path %d", mode);
  h->host_path = mode;
  return NEO_MPC_OK;
}

// Page-lock / release a host array of the caller's (hipHostRegister / hipHostUnregister behind a C signature, for
// callers built without the HIP headers -- the nav2 plugin is plain g++).
int neo_mpc_pin_host_memory(void* ptr, size_t bytes) {
  if (!ptr || bytes == 0) return fail(NEO_MPC_ERR_INVALID_ARGUMENT, "null argument");
  HIP_TRY(hipHostRegister(ptr, bytes, hipHostRegisterMapped | hipHostRegisterPortable));
  return NEO_MPC_OK;
}
int neo_mpc_unpin_host_memory(void* ptr) {
  if (!ptr) return fail(NEO_MPC_ERR_INVALID_ARGUMENT, "null argument");
  HIP_TRY(hipHostUnregister(ptr));
  return NEO_MPC_OK;
}

int neo_mpc_postprocess_batch(neo_mpc_handle* h, const neo_mpc_batch* batch, const int32_t* success) {
  SolveArgs a;
  int rc = fill_args(h, batch, a);
  if (rc) return rc;
  if (!batch->solution) return fail(NEO_MPC_ERR_INVALID_ARGUMENT, "postprocess needs batch->solution (x.x)");
  if (batch->count == 0) return NEO_MPC_OK;
  HIP_TRY(hipSetDevice(h->device));
  neo_mpc_batch d;
  auto bail = [](int code) { (void)hipStreamSynchronize(nullptr); return code; };
  if ((rc = stage_in(h, batch, d, true))) return bail(rc);
  if ((rc = fill_args(h, &d, a))) return bail(rc);
  if (success) {
    if ((rc = h->success.reserve(batch->count * 4))) return bail(rc);
    if (hipMemcpy(h->success.ptr, success, batch->count * 4, hipMemcpyHostToDevice) != hipSuccess)
      return bail(fail(NEO_MPC_ERR_DEVICE, "copy of the success flags failed"));
    a.success = (const int32_t*)h->success.ptr;
  }
  if ((rc = map_acquire(h, nullptr))) return bail(rc);
  launch_postprocess(a, nullptr);
  if (hipGetLastError() != hipSuccess) return bail(fail(NEO_MPC_ERR_DEVICE, "kernel launch failed"));
  if ((rc = map_release(h, nullptr))) return bail(rc);
  return bail(stage_out(h, batch, false));
}

int neo_mpc_objective_batch(neo_mpc_handle* h, const neo_mpc_problem* problems, const double* u, double* cost_out,
                            size_t count) {
  if (!h || !problems || !u || !cost_out) return fail(NEO_MPC_ERR_INVALID_ARGUMENT, "null argument");
  if (!h->has_map) return fail(NEO_MPC_ERR_NO_COSTMAP, "neo_mpc_set_costmap has not been called");
  if (count == 0) return NEO_MPC_OK;
  HIP_TRY(hipSetDevice(h->device));
  const size_t nv = 3 * (size_t)h->params.control_steps;
  int rc;
  if ((rc = h->problems.reserve(count * sizeof(neo_mpc_problem)))) return rc;
  if ((rc = h->u.reserve(count * nv * 8))) return rc;
  if ((rc = h->cost.reserve(count * 8))) return rc;
  HIP_TRY(hipMemcpy(h->problems.ptr, problems, count * sizeof(neo_mpc_problem), hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(h->u.ptr, u, count * nv * 8, hipMemcpyHostToDevice));
  ObjectiveArgs a;
  std::memset(&a, 0, sizeof(a));
  a.problems = (const neo_mpc_problem*)h->problems.ptr;
  a.u = (const double*)h->u.ptr;
  a.cost = (double*)h->cost.ptr;
  a.term_table = (const double*)h->term_buf.ptr;
  a.count = (uint32_t)count;
  a.p = h->dp; a.map = h->map;
  a.w_trans = h->params.w_trans; a.w_orient = h->params.w_orient; a.w_control = h->params.w_control;
  a.w_terminal = h->params.w_terminal; a.w_costmap = h->params.w_costmap;
  if ((rc = map_acquire(h, nullptr))) return rc;
  launch_objective(a, nullptr);
  HIP_TRY(hipGetLastError());
  if ((rc = map_release(h, nullptr))) return rc;
  HIP_TRY(hipMemcpy(cost_out, h->cost.ptr, count * 8, hipMemcpyDeviceToHost));
  return NEO_MPC_OK;
}

static int hook_batch(neo_mpc_handle* h, const neo_mpc_problem* problems, const double* u, double* out, size_t count,
                      int max_it);
int neo_mpc_gradient_batch(neo_mpc_handle* h, const neo_mpc_problem* problems, const double* u, double* grad_out,
                           size_t count) {
  return hook_batch(h, problems, u, grad_out, count, kDumpGradient);
}
int neo_mpc_direction_batch(neo_mpc_handle* h, const neo_mpc_problem* problems, const double* u, double* dir_out,
                            size_t count, int iteration) {
  if (iteration < 0 || iteration > 1000) return fail(NEO_MPC_ERR_INVALID_ARGUMENT, "iteration %d", iteration);
  return hook_batch(h, problems, u, dir_out, count, kDumpGradient + 1 + iteration);
}
static int hook_batch(neo_mpc_handle* h, const neo_mpc_problem* problems, const double* u, double* grad_out, size_t count,
                      int max_it) {
  if (!h || !problems || !u || !grad_out) return fail(NEO_MPC_ERR_INVALID_ARGUMENT, "null argument");
  if (!h->has_map) return fail(NEO_MPC_ERR_NO_COSTMAP, "neo_mpc_set_costmap has not been called");
  if (count == 0) return NEO_MPC_OK;
  if (count > 0x7fffffffull) return fail(NEO_MPC_ERR_INVALID_ARGUMENT, "count too large");
  HIP_TRY(hipSetDevice(h->device));
  const size_t nv = 3 * (size_t)h->params.control_steps;
  // K1 itself with `u` as the warm start of instances whose goal is unchanged (no reset), stopped right
  // after its first gradient pass (DevParams.max_it = kDumpGradient): the gradient comes out of `solution`
  std::vector<neo_mpc_state> st(count);
  std::memset(st.data(), 0, count * sizeof(neo_mpc_state));
  for (size_t i = 0; i < count; ++i) {
    for (int k = 0; k < 3; ++k) st[i].old_goal[k] = problems[i].goal_xyz[k];
    for (int k = 0; k < 4; ++k) st[i].old_goal[3 + k] = problems[i].goal_q[k];
    st[i].has_old_goal = 1;
  }
  int rc;
  if ((rc = h->problems.reserve(count * sizeof(neo_mpc_problem)))) return rc;
  if ((rc = h->states.reserve(count * sizeof(neo_mpc_state)))) return rc;
  if ((rc = h->warm.reserve(count * nv * 8))) return rc;
  if ((rc = h->commands.reserve(count * sizeof(neo_mpc_command)))) return rc;
  if ((rc = h->solution.reserve(count * nv * 8))) return rc;
  HIP_TRY(hipMemcpy(h->problems.ptr, problems, count * sizeof(neo_mpc_problem), hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(h->states.ptr, st.data(), count * sizeof(neo_mpc_state), hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(h->warm.ptr, u, count * nv * 8, hipMemcpyHostToDevice));
  neo_mpc_batch d;
  std::memset(&d, 0, sizeof(d));
  d.count = count;
  d.problems = (const neo_mpc_problem*)h->problems.ptr;
  d.states = (neo_mpc_state*)h->states.ptr;
  d.warm_start = (double*)h->warm.ptr;
  d.commands = (neo_mpc_command*)h->commands.ptr;
  d.solution = (double*)h->solution.ptr;
  SolveArgs a;
  if ((rc = fill_args(h, &d, a))) return rc;
  a.p.max_it = max_it;
  HIP_TRY(hipMemset(h->solution.ptr, 0xFF, count * nv * 8));   // NaN rows for instances that stop before the dump
  if ((rc = map_acquire(h, nullptr))) return rc;
  launch_solve(a, h->tuning, nullptr);
  HIP_TRY(hipGetLastError());
  if ((rc = map_release(h, nullptr))) return rc;
  HIP_TRY(hipMemcpy(grad_out, h->solution.ptr, count * nv * 8, hipMemcpyDeviceToHost));
  return NEO_MPC_OK;
}

static int check_plan_batch(const neo_mpc_handle* h, const neo_mpc_lookahead_params* lp, const neo_mpc_plan_batch* b) {
  if (!h || !lp || !b) return fail(NEO_MPC_ERR_INVALID_ARGUMENT, "null argument");
  if (b->count > 0 && (!b->plan_poses || !b->plan_offsets || !b->robot_poses || !b->slow_down || !b->carrots))
    return fail(NEO_MPC_ERR_INVALID_ARGUMENT, "plan_poses/plan_offsets/robot_poses/slow_down/carrots must not be null");
  if (b->count > 0x7fffffffull) return fail(NEO_MPC_ERR_INVALID_ARGUMENT, "count too large");
  return NEO_MPC_OK;
}

int neo_mpc_select_carrots_device(neo_mpc_handle* h, const neo_mpc_lookahead_params* lp,
                                  const neo_mpc_plan_batch* b, void* stream) {
  int rc = check_plan_batch(h, lp, b);
  if (rc) return rc;
  CarrotArgs a;
  a.lp = *lp;
  a.b = *b;
  HIP_TRY(hipSetDevice(h->device));
  launch_carrots(a, stream);
  HIP_TRY(hipGetLastError());
  return NEO_MPC_OK;
}

int neo_mpc_select_carrots(neo_mpc_handle* h, const neo_mpc_lookahead_params* lp, const neo_mpc_plan_batch* b) {
  int rc = check_plan_batch(h, lp, b);
  if (rc) return rc;
  if (b->count == 0) return NEO_MPC_OK;
  HIP_TRY(hipSetDevice(h->device));
  const size_t n = b->count;
  const size_t total = b->plan_offsets[n];
  if ((rc = h->plan_poses.reserve(total * 24 + 8))) return rc;
  if ((rc = h->plan_offsets.reserve((n + 1) * 4))) return rc;
  if ((rc = h->robot_poses.reserve(n * 24))) return rc;
  if ((rc = h->fp_costs.reserve(n * 8))) return rc;
  if ((rc = h->slow_down.reserve(n * 4))) return rc;
  if ((rc = h->carrots.reserve(n * sizeof(neo_mpc_carrot)))) return rc;
  HIP_TRY(hipMemcpy(h->plan_poses.ptr, b->plan_poses, total * 24, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(h->plan_offsets.ptr, b->plan_offsets, (n + 1) * 4, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(h->robot_poses.ptr, b->robot_poses, n * 24, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(h->slow_down.ptr, b->slow_down, n * 4, hipMemcpyHostToDevice));
  CarrotArgs a;
  a.lp = *lp;
  a.b = *b;
  a.b.plan_poses = (const double*)h->plan_poses.ptr;
  a.b.plan_offsets = (const uint32_t*)h->plan_offsets.ptr;
  a.b.robot_poses = (const double*)h->robot_poses.ptr;
  a.b.slow_down = (int32_t*)h->slow_down.ptr;
  a.b.carrots = (neo_mpc_carrot*)h->carrots.ptr;
  a.b.footprint_costs = nullptr;
  if (b->footprint_costs) {
    HIP_TRY(hipMemcpy(h->fp_costs.ptr, b->footprint_costs, n * 8, hipMemcpyHostToDevice));
    a.b.footprint_costs = (const double*)h->fp_costs.ptr;
  }
  a.b.problems = nullptr;
  if (b->problems) {
    if ((rc = h->problems.reserve(n * sizeof(neo_mpc_problem)))) return rc;
    HIP_TRY(hipMemcpy(h->problems.ptr, b->problems, n * sizeof(neo_mpc_problem), hipMemcpyHostToDevice));
    a.b.problems = (neo_mpc_problem*)h->problems.ptr;
  }
  launch_carrots(a, nullptr);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpy(b->carrots, h->carrots.ptr, n * sizeof(neo_mpc_carrot), hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(b->slow_down, h->slow_down.ptr, n * 4, hipMemcpyDeviceToHost));
  if (b->problems)
    HIP_TRY(hipMemcpy(b->problems, h->problems.ptr, n * sizeof(neo_mpc_problem), hipMemcpyDeviceToHost));
  return NEO_MPC_OK;
}

int neo_mpc_kernel_info(const neo_mpc_handle* h, uint32_t* lds_bytes, uint32_t* reach_cells, uint32_t* tile_in_lds) {
  if (!h) return fail(NEO_MPC_ERR_INVALID_ARGUMENT, "null handle");
  if (lds_bytes) *lds_bytes = (uint32_t)h->lds.total_bytes;
  if (reach_cells) *reach_cells = (uint32_t)h->lds.reach;
  if (tile_in_lds) *tile_in_lds = h->lds.tile_w ? 1u : 0u;
  return NEO_MPC_OK;
}

}  // extern "C"
