// plugin_seam.cpp -- what a maintainer of neobotix/neo_mpc_planner2 writes in
// src/NeoMpcPlanner.cpp to call the HIP solver in-process instead of the ROS2 service hop.
//
// It replaces exactly the seam at cpp:240-252 (build Optimizer::Request, async_send_request,
// blocking result.get()) and the client set-up at cpp:308, 325-330; everything else of the
// nav2_core::Controller plugin (plan pruning, look-ahead, footprint gate, lifecycle, YAML,
// pluginlib export) stays as it is.  ROS2 / nav2 are not installed in this image, so the
// message types below are minimal stand-ins with the fields the seam touches; with the real
// headers the struct definitions disappear and the two functions are pasted as shown.
//
//   g++ -std=c++17 -I include examples/plugin_seam.cpp -L neo_mpc_planner2_amd -lneo_mpc
//       (tests/test_abi.py::test_plugin_seam_example_compiles_and_links does this)
#include <chrono>
#include <functional>
#include <stdexcept>
#include <string>

#include "neo_mpc.h"

// ---- stand-ins for geometry_msgs / nav2_costmap_2d (only the members the seam uses) ----------
namespace geometry_msgs::msg {
struct Point { double x = 0, y = 0, z = 0; };
struct Quaternion { double x = 0, y = 0, z = 0, w = 1; };
struct Pose { Point position; Quaternion orientation; };
struct PoseStamped { Pose pose; };
struct Vector3 { double x = 0, y = 0, z = 0; };
struct Twist { Vector3 linear, angular; };
struct TwistStamped { Twist twist; };
}  // namespace geometry_msgs::msg
namespace nav2_costmap_2d {
struct Costmap2D {  // getCharMap()/getSizeInCells*/getResolution/getOrigin* as in nav2
  unsigned char* getCharMap() const { return cells; }
  unsigned int getSizeInCellsX() const { return sx; }
  unsigned int getSizeInCellsY() const { return sy; }
  double getResolution() const { return res; }
  double getOriginX() const { return ox; }
  double getOriginY() const { return oy; }
  unsigned char* cells = nullptr; unsigned int sx = 0, sy = 0; double res = 0.05, ox = 0, oy = 0;
};
}  // namespace nav2_costmap_2d
struct ControllerException : std::runtime_error { using std::runtime_error::runtime_error; };

// ---- members added to class NeoMpcPlanner (include/NeoMpcPlanner.h) -----------------------
struct NeoMpcPlannerSeam {
  neo_mpc_handle* mpc_ = nullptr;       // replaces rclcpp::Client<neo_srvs2::srv::Optimizer>::SharedPtr client
  neo_mpc_state mpc_state_{};           // the state the Python node kept (py:115-152)
  double mpc_warm_[3 * NEO_MPC_MAX_CONTROL_STEPS] = {};
  double mpc_last_call_ = 0.0;           // seconds on mpc_clock_
  bool mpc_called_ = false;
  // the Python node stamped its calls with time.time() (py:369); a test injects a deterministic clock here
  std::function<double()> mpc_clock_ = [] {
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
  };
  double control_frequency = 30.0;
  bool closer_to_goal = false;
  bool withhold_hint_ = false;          // (test: a caller that rebuilds neo_mpc_state from the node's attributes every tick
                                        // loses the build's own hint, has_prev_u0 / prev_u0 -- results must hold without it)
  neo_mpc_problem last_request_{};      // (diagnostics: what the last tick sent and got -- `--dump` records them)
  neo_mpc_command last_command_{};

  // configure(): replaces create_client (cpp:308) and the wait_for_service loop (cpp:325-330).
  // `declare` stands for the plugin's declare_parameter_if_not_declared/get_parameter pair; the
  // parameter names are the Python node's (py:49-75), now read from the plugin's namespace.
  template <class Declare>
  void configureSolver(Declare declare) {
    neo_mpc_params p;
    neo_mpc_default_params(&p);
    p.acc_x_limit = declare("acc_x_limit", p.acc_x_limit);
    p.acc_y_limit = declare("acc_y_limit", p.acc_y_limit);
    p.acc_theta_limit = declare("acc_theta_limit", p.acc_theta_limit);
    p.min_vel_x = declare("min_vel_x", p.min_vel_x);
    p.min_vel_y = declare("min_vel_y", p.min_vel_y);
    p.min_vel_theta = declare("min_vel_theta", p.min_vel_theta);
    p.max_vel_x = declare("max_vel_x", p.max_vel_x);
    p.max_vel_y = declare("max_vel_y", p.max_vel_y);
    p.max_vel_trans = declare("max_vel_trans", p.max_vel_trans);
    p.max_vel_theta = declare("max_vel_theta", p.max_vel_theta);
    p.w_trans = declare("w_trans", p.w_trans);
    p.w_orient = declare("w_orient", p.w_orient);
    p.w_control = declare("w_control", p.w_control);
    p.w_terminal = declare("w_terminal", p.w_terminal);
    p.w_costmap = declare("w_costmap", p.w_costmap);
    p.w_footprint = declare("w_footprint", p.w_footprint);
    p.low_pass_gain = declare("low_pass_gain", p.low_pass_gain);
    p.opt_tolerance = declare("opt_tolerance", p.opt_tolerance);
    p.prediction_horizon = declare("prediction_horizon", p.prediction_horizon);
    p.control_steps = (int32_t)declare("control_steps", (double)p.control_steps);
    mpc_ = neo_mpc_create(&p, /*device=*/0);
    if (!mpc_) throw ControllerException(std::string("neo_mpc_create: ") + neo_mpc_last_error());
    mpc_state_ = neo_mpc_state{};
    mpc_state_.waiting_time = p.waiting_time;
  }

  void cleanupSolver() { neo_mpc_destroy(mpc_); mpc_ = nullptr; }

  // computeVelocityCommands(): the body that replaces cpp:240-252.
  geometry_msgs::msg::TwistStamped solve(const geometry_msgs::msg::PoseStamped& position,
                                         const geometry_msgs::msg::Twist& speed,
                                         const geometry_msgs::msg::PoseStamped& carrot_pose,
                                         const geometry_msgs::msg::Pose& goal_pose,
                                         const nav2_costmap_2d::Costmap2D& costmap, double footprint_cost_raw) {
    // the Python node read the local costmap from a topic; here it is handed over each tick
    if (neo_mpc_set_costmap(mpc_, costmap.getCharMap(), costmap.getSizeInCellsX(), costmap.getSizeInCellsY(),
                            costmap.getResolution(), costmap.getOriginX(), costmap.getOriginY()) != NEO_MPC_OK)
      throw ControllerException(neo_mpc_last_error());

    neo_mpc_problem req{};                                   // == Optimizer::Request (cpp:240-246)
    req.cur_xy[0] = position.pose.position.x;                // request->current_pose = position
    req.cur_xy[1] = position.pose.position.y;
    req.cur_q[0] = position.pose.orientation.x; req.cur_q[1] = position.pose.orientation.y;
    req.cur_q[2] = position.pose.orientation.z; req.cur_q[3] = position.pose.orientation.w;
    req.carrot_xy[0] = carrot_pose.pose.position.x;          // request->carrot_pose = carrot_pose
    req.carrot_xy[1] = carrot_pose.pose.position.y;
    req.carrot_q[0] = carrot_pose.pose.orientation.x; req.carrot_q[1] = carrot_pose.pose.orientation.y;
    req.carrot_q[2] = carrot_pose.pose.orientation.z; req.carrot_q[3] = carrot_pose.pose.orientation.w;
    req.goal_xyz[0] = goal_pose.position.x; req.goal_xyz[1] = goal_pose.position.y;   // request->goal_pose
    req.goal_xyz[2] = goal_pose.position.z;
    req.goal_q[0] = goal_pose.orientation.x; req.goal_q[1] = goal_pose.orientation.y;
    req.goal_q[2] = goal_pose.orientation.z; req.goal_q[3] = goal_pose.orientation.w;
    req.cur_vel[0] = speed.linear.x; req.cur_vel[1] = speed.linear.y; req.cur_vel[2] = speed.angular.z;
    req.control_interval = 1.0 / control_frequency;          // cpp:246
    const double now = mpc_clock_();                          // py:369-371 (wall clock between calls)
    req.delta_t = mpc_called_ ? now - mpc_last_call_ : 1.0e9;
    mpc_last_call_ = now; mpc_called_ = true;
    // footprintCostAtPose (cpp:218-219) is already computed by the plugin on nav2's 0..255 scale;
    // 254 (lethal) is what the Python node saw as getFootprintCost(...) == 1.0 (py:343)
    req.footprint_cost = footprint_cost_raw >= 254.0 ? 1.0 : 0.0;
    req.switch_opt = closer_to_goal ? 1 : 0;   // cpp:245 request->switch_opt = closer_to_goal (stored by the node, py:354, never read)

    if (withhold_hint_) { mpc_state_.has_prev_u0 = 0; mpc_state_.prev_u0[0] = mpc_state_.prev_u0[1] = mpc_state_.prev_u0[2] = 0.0; }
    last_request_ = req;
    neo_mpc_command out{};
    neo_mpc_batch batch{};
    batch.count = 1;
    batch.problems = &req; batch.states = &mpc_state_; batch.warm_start = mpc_warm_; batch.commands = &out;
    if (neo_mpc_solve_batch(mpc_, &batch) != NEO_MPC_OK)     // == async_send_request + result.get()
      throw ControllerException(neo_mpc_last_error());

    last_command_ = out;
    geometry_msgs::msg::TwistStamped cmd_vel_final;           // == out->output_vel (cpp:251-252)
    cmd_vel_final.twist.linear.x = out.vel[0];
    cmd_vel_final.twist.linear.y = out.vel[1];
    cmd_vel_final.twist.angular.z = out.vel[2];
    return cmd_vel_final;
  }
};

// `plugin_seam` alone is the link check (no GPU needed).  `plugin_seam --run` drives the seam the way
// nav2's controller server would: 300 control ticks at 30 Hz of one robot chasing a look-ahead point
// that slides along a straight plan, costmap handed over every tick, README parameters
// (README.md:53-84) -- and prints the per-tick latency of `solve()` (set_costmap + solve_batch,
// host buffers, count = 1).  tests/test_gpu_parity.py runs it on the GPU box.
//   --ticks N        number of control ticks (default 300)
//   --fake-clock     the wall clock of py:369 advances exactly one control interval per tick (deterministic latch)
//   --obstacle       the rolling costmap CHANGES while the robot drives: a lethal block appears across the plan at
//                    tick 60 and is gone again at tick 260 -- the per-tick costmap hand-over (cpp:290-334's
//                    costmap_, getCharMap()) is what makes the robot stop in front of it (collision latch,
//                    py:312-347, 374-382) and move on afterwards
//   --withhold-hint  the state record is handed over without the build's own hint (has_prev_u0 = 0) every tick
//   --dump FILE      every tick's request, state and warm start before and after, command and costmap version
//                    (tests replay them through the oracle, tick by tick)
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

int main(int argc, char** argv) {
  if (argc < 2 || std::strcmp(argv[1], "--run") != 0) {
    NeoMpcPlannerSeam seam;
    (void)seam;
    return neo_mpc_abi_version() == NEO_MPC_ABI_VERSION ? 0 : 1;
  }
  int ticks = 300;
  bool fake_clock = false, obstacle = false, withhold_hint = false;
  const char* dump_path = nullptr;
  for (int k = 2; k < argc; ++k) {
    if (!std::strcmp(argv[k], "--ticks") && k + 1 < argc) ticks = std::atoi(argv[++k]);
    else if (!std::strcmp(argv[k], "--fake-clock")) fake_clock = true;
    else if (!std::strcmp(argv[k], "--obstacle")) obstacle = true;
    else if (!std::strcmp(argv[k], "--dump") && k + 1 < argc) dump_path = argv[++k];
    else if (!std::strcmp(argv[k], "--withhold-hint")) withhold_hint = true;
    else { std::fprintf(stderr, "unknown option %s\n", argv[k]); return 64; }
  }
  NeoMpcPlannerSeam seam;
  seam.withhold_hint_ = withhold_hint;
  seam.configureSolver([](const char* name, double dflt) {   // the README's YAML block
    const struct { const char* n; double v; } readme[] = {
        {"acc_x_limit", 2.5}, {"acc_y_limit", 2.5}, {"acc_theta_limit", 3.0}, {"min_vel_x", -0.7}, {"min_vel_y", -0.7},
        {"min_vel_theta", -0.7}, {"max_vel_x", 0.7}, {"max_vel_y", 0.7}, {"max_vel_trans", 0.7}, {"max_vel_theta", 0.7},
        {"w_trans", 0.82}, {"w_orient", 0.50}, {"w_control", 0.05}, {"w_terminal", 0.05}, {"w_footprint", 0.0},
        {"w_costmap", 0.05}, {"low_pass_gain", 0.5}, {"opt_tolerance", 1e-3}, {"prediction_horizon", 0.8},
        {"control_steps", 3.0}};
    for (const auto& kv : readme) if (std::strcmp(kv.n, name) == 0) return kv.v;
    return dflt;
  });
  double fake_now = 1000.0;
  if (fake_clock) seam.mpc_clock_ = [&fake_now] { return fake_now; };
  const unsigned S = 200;                                      // 10 m x 10 m rolling window, 5 cm cells
  std::vector<unsigned char> cells(S * S, 0);
  for (unsigned y = 120; y < 130; ++y) for (unsigned x = 40; x < 160; ++x) cells[y * S + x] = 254;   // a wall north of the path
  const std::vector<unsigned char> cells_clear = cells;
  std::vector<unsigned char> cells_blocked = cells;
  for (unsigned y = 70; y < 120; ++y) for (unsigned x = 110; x < 114; ++x) cells_blocked[y * S + x] = 254;   // across the plan at x = 0.5 m
  nav2_costmap_2d::Costmap2D costmap;
  costmap.cells = cells.data(); costmap.sx = S; costmap.sy = S; costmap.res = 0.05; costmap.ox = -5.0; costmap.oy = -5.0;
  std::FILE* dump = nullptr;
  neo_mpc_params used{};
  neo_mpc_get_params(seam.mpc_, &used);
  const int nv = 3 * used.control_steps;
  if (dump_path) {
    dump = std::fopen(dump_path, "wb");
    if (!dump) { std::perror(dump_path); return 65; }
    // header: magic, ticks, control_steps, map size, maps; parameters; map geometry; the two costmap versions
    const char magic[8] = {'N', 'E', 'O', 'S', 'E', 'A', 'M', '1'};
    const int32_t hdr[4] = {ticks, used.control_steps, (int32_t)S, 2};
    const double geom[3] = {costmap.res, costmap.ox, costmap.oy};
    std::fwrite(magic, 1, 8, dump); std::fwrite(hdr, 4, 4, dump); std::fwrite(&used, sizeof(used), 1, dump);
    std::fwrite(geom, 8, 3, dump);
    std::fwrite(cells_clear.data(), 1, cells_clear.size(), dump);
    std::fwrite(cells_blocked.data(), 1, cells_blocked.size(), dump);
  }
  double x = -3.0, y = 0.0, yaw = 0.3;
  geometry_msgs::msg::Twist speed;
  geometry_msgs::msg::Pose goal;
  goal.position.x = 4.0; goal.orientation.w = 1.0;
  std::vector<double> us;
  double worst = 0.0;
  int stopped_ticks = 0;
  for (int tick = 0; tick < ticks; ++tick) {
    // the local costmap as nav2 would have updated it by now, in the buffer getCharMap() points at
    const bool blocked = obstacle && tick >= 60 && tick < 260;
    std::memcpy(cells.data(), blocked ? cells_blocked.data() : cells_clear.data(), cells.size());
    geometry_msgs::msg::PoseStamped position, carrot;
    position.pose.position.x = x; position.pose.position.y = y;
    position.pose.orientation.z = std::sin(0.5 * yaw); position.pose.orientation.w = std::cos(0.5 * yaw);
    // look-ahead point 0.4 m further along the plan y = 0, heading 0, expressed in the base frame
    const double cxw = std::min(x + 0.4, 4.0), cyw = 0.0;
    const double dx = cxw - x, dy = cyw - y;
    carrot.pose.position.x = std::cos(yaw) * dx + std::sin(yaw) * dy;
    carrot.pose.position.y = -std::sin(yaw) * dx + std::cos(yaw) * dy;
    carrot.pose.orientation.z = std::sin(-0.5 * yaw); carrot.pose.orientation.w = std::cos(-0.5 * yaw);
    neo_mpc_state state_before = seam.mpc_state_;   // (what the solver is handed: without the hint when it is withheld)
    if (seam.withhold_hint_) { state_before.has_prev_u0 = 0; state_before.prev_u0[0] = state_before.prev_u0[1] = state_before.prev_u0[2] = 0.0; }
    double warm_before[3 * NEO_MPC_MAX_CONTROL_STEPS];
    std::memcpy(warm_before, seam.mpc_warm_, sizeof(warm_before));
    fake_now += 1.0 / seam.control_frequency;
    const auto t0 = std::chrono::steady_clock::now();
    const auto cmd = seam.solve(position, speed, carrot, goal, costmap, /*footprint_cost_raw=*/0.0);
    const double dt_us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    if (tick >= 20) us.push_back(dt_us);
    if (dump) {
      const int32_t which[2] = {blocked ? 1 : 0, tick};
      std::fwrite(which, 4, 2, dump);
      std::fwrite(&seam.last_request_, sizeof(neo_mpc_problem), 1, dump);
      std::fwrite(&state_before, sizeof(neo_mpc_state), 1, dump);
      std::fwrite(warm_before, 8, nv, dump);
      std::fwrite(&seam.last_command_, sizeof(neo_mpc_command), 1, dump);
      std::fwrite(&seam.mpc_state_, sizeof(neo_mpc_state), 1, dump);
      std::fwrite(seam.mpc_warm_, 8, nv, dump);
    }
    const double vx = cmd.twist.linear.x, vy = cmd.twist.linear.y, w = cmd.twist.angular.z;
    if (!std::isfinite(vx) || !std::isfinite(vy) || !std::isfinite(w)) { std::printf("non-finite command at tick %d\n", tick); return 2; }
    if (seam.last_command_.flags & NEO_MPC_FLAG_STOPPED) ++stopped_ticks;
    worst = std::max(worst, std::hypot(vx, vy));
    yaw += w / 30.0;
    x += (vx * std::cos(yaw) - vy * std::sin(yaw)) / 30.0;
    y += (vx * std::sin(yaw) + vy * std::cos(yaw)) / 30.0;
    speed.linear.x = vx; speed.linear.y = vy; speed.angular.z = w;
  }
  if (dump) std::fclose(dump);
  seam.cleanupSolver();
  std::sort(us.begin(), us.end());
  std::printf("{\"what\": \"plugin seam (C++ -> C-ABI, host buffers, count = 1), %d warm ticks\", \"tick_us_median\": %.1f, "
              "\"tick_us_p99\": %.1f, \"final_x\": %.3f, \"final_y\": %.3f, \"final_yaw\": %.3f, \"max_speed\": %.3f, "
              "\"stopped_ticks\": %d}\n",
              (int)us.size(), us[us.size() / 2], us[us.size() * 99 / 100], x, y, yaw, worst, stopped_ticks);
  return (x > 3.0 && std::fabs(y) < 0.2 && worst <= 0.7 + 1e-9) ? 0 : 3;
}
