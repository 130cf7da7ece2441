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
  std::chrono::steady_clock::time_point mpc_last_call_{};
  bool mpc_called_ = false;
  double control_frequency = 30.0;
  bool closer_to_goal = false;

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
    const auto now = std::chrono::steady_clock::now();        // py:369-371 (wall clock between calls)
    req.delta_t = mpc_called_ ? std::chrono::duration<double>(now - mpc_last_call_).count() : 1.0e9;
    mpc_last_call_ = now; mpc_called_ = true;
    // footprintCostAtPose (cpp:218-219) is already computed by the plugin on nav2's 0..255 scale;
    // 254 (lethal) is what the Python node saw as getFootprintCost(...) == 1.0 (py:343)
    req.footprint_cost = footprint_cost_raw >= 254.0 ? 1.0 : 0.0;
    req.switch_opt = 0;                        // cpp:245 closer_to_goal (stored by the node, py:354, never read)

    neo_mpc_command out{};
    neo_mpc_batch batch{};
    batch.count = 1;
    batch.problems = &req; batch.states = &mpc_state_; batch.warm_start = mpc_warm_; batch.commands = &out;
    if (neo_mpc_solve_batch(mpc_, &batch) != NEO_MPC_OK)     // == async_send_request + result.get()
      throw ControllerException(neo_mpc_last_error());

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
  NeoMpcPlannerSeam seam;
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
  const unsigned S = 200;                                      // 10 m x 10 m rolling window, 5 cm cells
  std::vector<unsigned char> cells(S * S, 0);
  for (unsigned y = 120; y < 130; ++y) for (unsigned x = 40; x < 160; ++x) cells[y * S + x] = 254;   // a wall north of the path
  nav2_costmap_2d::Costmap2D costmap;
  costmap.cells = cells.data(); costmap.sx = S; costmap.sy = S; costmap.res = 0.05; costmap.ox = -5.0; costmap.oy = -5.0;
  double x = -3.0, y = 0.0, yaw = 0.3;
  geometry_msgs::msg::Twist speed;
  geometry_msgs::msg::Pose goal;
  goal.position.x = 4.0; goal.orientation.w = 1.0;
  std::vector<double> us;
  double worst = 0.0;
  for (int tick = 0; tick < 300; ++tick) {
    geometry_msgs::msg::PoseStamped position, carrot;
    position.pose.position.x = x; position.pose.position.y = y;
    position.pose.orientation.z = std::sin(0.5 * yaw); position.pose.orientation.w = std::cos(0.5 * yaw);
    // look-ahead point 0.4 m further along the plan y = 0, heading 0, expressed in the base frame
    const double cxw = std::min(x + 0.4, 4.0), cyw = 0.0;
    const double dx = cxw - x, dy = cyw - y;
    carrot.pose.position.x = std::cos(yaw) * dx + std::sin(yaw) * dy;
    carrot.pose.position.y = -std::sin(yaw) * dx + std::cos(yaw) * dy;
    carrot.pose.orientation.z = std::sin(-0.5 * yaw); carrot.pose.orientation.w = std::cos(-0.5 * yaw);
    const auto t0 = std::chrono::steady_clock::now();
    const auto cmd = seam.solve(position, speed, carrot, goal, costmap, /*footprint_cost_raw=*/0.0);
    const double dt_us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    if (tick >= 20) us.push_back(dt_us);
    const double vx = cmd.twist.linear.x, vy = cmd.twist.linear.y, w = cmd.twist.angular.z;
    if (!std::isfinite(vx) || !std::isfinite(vy) || !std::isfinite(w)) { std::printf("non-finite command at tick %d\n", tick); return 2; }
    worst = std::max(worst, std::hypot(vx, vy));
    yaw += w / 30.0;
    x += (vx * std::cos(yaw) - vy * std::sin(yaw)) / 30.0;
    y += (vx * std::sin(yaw) + vy * std::cos(yaw)) / 30.0;
    speed.linear.x = vx; speed.linear.y = vy; speed.angular.z = w;
  }
  seam.cleanupSolver();
  std::sort(us.begin(), us.end());
  std::printf("{\"what\": \"plugin seam (C++ -> C-ABI, host buffers, count = 1), 280 warm ticks\", \"tick_us_median\": %.1f, "
              "\"tick_us_p99\": %.1f, \"final_x\": %.3f, \"final_y\": %.3f, \"final_yaw\": %.3f, \"max_speed\": %.3f}\n",
              us[us.size() / 2], us[us.size() * 99 / 100], x, y, yaw, worst);
  return (x > 3.0 && std::fabs(y) < 0.2 && worst <= 0.7 + 1e-9) ? 0 : 3;
}
