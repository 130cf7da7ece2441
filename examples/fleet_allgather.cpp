// fleet_allgather.cpp -- a multi-GPU fleet tick through the C-ABI alone (no Python, no torch):
// one process drives every visible GPU -- one neo_mpc_handle, one HIP stream and one RCCL communicator per
// device -- each GPU solves its contiguous block of the fleet's instances, and ONE all-gather of the packed
// (vx, vy, omega) commands puts every robot's command on every GPU (include/neo_mpc.h
// "multi-GPU fleets").  Checks the gathered buffer against the per-GPU results and prints one JSON line.
//
//   hipcc -O2 -I include examples/fleet_allgather.cpp -L neo_mpc_planner2_amd -lneo_mpc \
//         -Wl,-rpath,$PWD/neo_mpc_planner2_amd -o fleet_allgather && ./fleet_allgather [instances per GPU] [ranks]
// `ranks` (default: the visible GPUs): with more ranks than GPUs, rank d runs on device d modulo the GPUs -- RCCL itself
// refuses two ranks on one device; the tests run eight logical ranks on their one GPU against a stand-in library
// (NEO_MPC_RCCL_LIBRARY, tests/standin_rccl) to execute this file's group bracketing, offsets and gathered layout.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "neo_mpc.h"

#define CHECK_HIP(e)                                                                      \
  do { hipError_t r_ = (e); if (r_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(r_)); return 2; } } while (0)
#define CHECK_MPC(e)                                                                      \
  do { int r_ = (e); if (r_ != NEO_MPC_OK) { std::fprintf(stderr, "%s: %s\n", #e, neo_mpc_last_error()); return 3; } } while (0)

struct Rank {
  neo_mpc_handle* h = nullptr;
  hipStream_t stream = nullptr;
  void* comm = nullptr;
  neo_mpc_problem* problems = nullptr;
  neo_mpc_state* states = nullptr;
  double* warm = nullptr;
  neo_mpc_command* commands = nullptr;
  double *vel = nullptr, *all = nullptr;
};

int main(int argc, char** argv) {
  const size_t count = argc > 1 ? (size_t)std::atol(argv[1]) : 4096;
  int ngpu = 0;
  CHECK_HIP(hipGetDeviceCount(&ngpu));
  if (ngpu < 1) { std::fprintf(stderr, "no HIP device\n"); return 1; }
  const int ndev = argc > 2 ? std::atoi(argv[2]) : ngpu;   // ranks; rank d runs on device d % ngpu
  if (ndev < 1 || ndev > 64) { std::fprintf(stderr, "ranks must be 1..64\n"); return 1; }
  if (!neo_mpc_rccl_available()) { std::fprintf(stderr, "RCCL not available: %s\n", neo_mpc_last_error()); return 4; }

  neo_mpc_params prm;
  CHECK_MPC(neo_mpc_default_params(&prm));   // then the README sample block (README.md:53-84)
  prm.acc_x_limit = prm.acc_y_limit = 2.5; prm.acc_theta_limit = 3.0;
  prm.min_vel_x = prm.min_vel_y = prm.min_vel_theta = -0.7; prm.min_vel_trans = -0.7;
  prm.max_vel_x = prm.max_vel_y = prm.max_vel_theta = prm.max_vel_trans = 0.7;
  prm.w_trans = 0.82; prm.w_orient = 0.5; prm.w_control = 0.05; prm.w_terminal = 0.05; prm.w_costmap = 0.05;
  prm.w_footprint = 0; prm.opt_tolerance = 1e-3; prm.prediction_horizon = 0.8; prm.control_steps = 3;

  // a 500 x 500 costmap with a few lethal blobs, replicated on every GPU
  const uint32_t S = 500;
  std::vector<uint8_t> cells((size_t)S * S, 0);
  for (int k = 0; k < 20; ++k) {
    const int cx = 40 + (k * 97) % 420, cy = 40 + (k * 211) % 420;
    for (int y = cy - 8; y <= cy + 8; ++y)
      for (int x = cx - 8; x <= cx + 8; ++x)
        if ((x - cx) * (x - cx) + (y - cy) * (y - cy) <= 64) cells[(size_t)y * S + x] = 254;
  }

  std::vector<int> devices(ndev);
  std::vector<void*> comms(ndev);
  for (int d = 0; d < ndev; ++d) devices[d] = d % ngpu;
  CHECK_MPC(neo_mpc_comm_init_all(ndev, devices.data(), comms.data()));

  std::vector<Rank> ranks(ndev);
  std::vector<neo_mpc_problem> hp(count);
  std::vector<neo_mpc_state> hs(count);
  for (int d = 0; d < ndev; ++d) {
    Rank& r = ranks[d];
    CHECK_HIP(hipSetDevice(devices[d]));
    r.comm = comms[d];
    r.h = neo_mpc_create(&prm, devices[d]);
    if (!r.h) { std::fprintf(stderr, "create: %s\n", neo_mpc_last_error()); return 3; }
    CHECK_MPC(neo_mpc_set_costmap(r.h, cells.data(), S, S, 0.05, -12.5, -12.5));
    CHECK_HIP(hipStreamCreateWithFlags(&r.stream, hipStreamNonBlocking));
    CHECK_HIP(hipMalloc(&r.problems, count * sizeof(neo_mpc_problem)));
    CHECK_HIP(hipMalloc(&r.states, count * sizeof(neo_mpc_state)));
    CHECK_HIP(hipMalloc(&r.warm, count * 9 * sizeof(double)));
    CHECK_HIP(hipMalloc(&r.commands, count * sizeof(neo_mpc_command)));
    CHECK_HIP(hipMalloc(&r.vel, count * 3 * sizeof(double)));
    CHECK_HIP(hipMalloc(&r.all, (size_t)ndev * count * 3 * sizeof(double)));
    // this GPU's block of the fleet: instance b of rank d is robot d * count + b
    std::memset(hp.data(), 0, count * sizeof(neo_mpc_problem));
    std::memset(hs.data(), 0, count * sizeof(neo_mpc_state));
    for (size_t b = 0; b < count; ++b) {
      const double t = 0.37 * (double)(d * count + b);
      neo_mpc_problem& q = hp[b];
      q.cur_xy[0] = 10.0 * std::sin(t); q.cur_xy[1] = 10.0 * std::cos(1.7 * t);
      q.cur_q[2] = std::sin(0.5 * t); q.cur_q[3] = std::cos(0.5 * t);
      q.carrot_xy[0] = 0.4 * std::cos(2.3 * t); q.carrot_xy[1] = 0.4 * std::sin(2.3 * t);
      q.carrot_q[2] = std::sin(0.3 * std::sin(t)); q.carrot_q[3] = std::cos(0.3 * std::sin(t));
      q.goal_xyz[0] = 5.0 * std::cos(t); q.goal_xyz[1] = 5.0 * std::sin(0.9 * t);
      q.goal_q[2] = std::sin(0.25 * t); q.goal_q[3] = std::cos(0.25 * t);
      q.cur_vel[0] = 0.3 * std::sin(3.1 * t); q.cur_vel[1] = 0.3 * std::cos(2.9 * t); q.cur_vel[2] = 0.2 * std::sin(1.3 * t);
      q.control_interval = 1.0 / 30.0; q.delta_t = 1.0 / 30.0;
      for (int k = 0; k < 3; ++k) hs[b].last_control[k] = q.cur_vel[k];
    }
    CHECK_HIP(hipMemcpy(r.problems, hp.data(), count * sizeof(neo_mpc_problem), hipMemcpyHostToDevice));
    CHECK_HIP(hipMemcpy(r.states, hs.data(), count * sizeof(neo_mpc_state), hipMemcpyHostToDevice));
    CHECK_HIP(hipMemset(r.warm, 0, count * 9 * sizeof(double)));
  }

  const int ticks = 20;
  double gather_ms = 0.0;
  auto t0 = std::chrono::steady_clock::now();
  for (int tick = 0; tick < ticks; ++tick) {
    for (int d = 0; d < ndev; ++d) {   // every GPU solves its block
      Rank& r = ranks[d];
      CHECK_HIP(hipSetDevice(devices[d]));
      neo_mpc_batch b;
      std::memset(&b, 0, sizeof(b));
      b.count = count; b.problems = r.problems; b.states = r.states; b.warm_start = r.warm; b.commands = r.commands;
      b.velocities = r.vel;
      CHECK_MPC(neo_mpc_solve_batch_device(r.h, &b, r.stream));
      // balanced dispatch: every 5th tick the robots are re-dealt over the SIMDs by their iteration counts (same stream,
      // behind the solve; no result depends on it)
      if (tick % 5 == 0) CHECK_MPC(neo_mpc_balance_dispatch_device(r.h, r.commands, count, r.stream));
    }
    auto g0 = std::chrono::steady_clock::now();
    CHECK_MPC(neo_mpc_group_start());   // the single exchange step
    for (int d = 0; d < ndev; ++d) {
      CHECK_HIP(hipSetDevice(devices[d]));
      CHECK_MPC(neo_mpc_allgather_velocities(ranks[d].vel, ranks[d].all, count, ranks[d].comm, ranks[d].stream));
    }
    CHECK_MPC(neo_mpc_group_end());
    for (int d = 0; d < ndev; ++d) { CHECK_HIP(hipSetDevice(devices[d])); CHECK_HIP(hipStreamSynchronize(ranks[d].stream)); }
    if (tick == ticks - 1) gather_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - g0).count();
  }
  const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();

  // every GPU must hold every robot's command: rank d's slice of `all` == rank d's own `vel`, on all GPUs
  bool ok = true;
  std::vector<double> own(count * 3), got((size_t)ndev * count * 3);
  for (int d = 0; d < ndev && ok; ++d) {
    CHECK_HIP(hipSetDevice(devices[d]));
    CHECK_HIP(hipMemcpy(got.data(), ranks[d].all, got.size() * sizeof(double), hipMemcpyDeviceToHost));
    for (int s = 0; s < ndev && ok; ++s) {
      CHECK_HIP(hipSetDevice(devices[s]));
      CHECK_HIP(hipMemcpy(own.data(), ranks[s].vel, own.size() * sizeof(double), hipMemcpyDeviceToHost));
      ok = std::memcmp(own.data(), got.data() + (size_t)s * count * 3, own.size() * sizeof(double)) == 0;
    }
  }
  double speed_max = 0.0;
  for (size_t b = 0; b < count; ++b) speed_max = std::fmax(speed_max, std::hypot(own[3 * b], own[3 * b + 1]));
  std::printf("{\"n_gpus\": %d, \"ranks\": %d, \"instances_per_gpu\": %zu, \"ticks\": %d, \"solves_per_s\": %.4g, \"last_gather_ms\": %.3f, "
              "\"gathered_equals_local\": %s, \"max_speed\": %.4f}\n",
              ngpu < ndev ? ngpu : ndev, ndev, count, ticks, (double)ndev * count * ticks / secs, gather_ms, ok ? "true" : "false", speed_max);
  for (int d = 0; d < ndev; ++d) {
    CHECK_HIP(hipSetDevice(devices[d]));
    neo_mpc_destroy(ranks[d].h);
    CHECK_MPC(neo_mpc_comm_destroy(ranks[d].comm));
  }
  return ok ? 0 : 5;
}
