/*
 * neo_mpc.h -- C-ABI of the MI355X-native batched MPC solver (libneo_mpc.so).
 *
 * Drop-in boundary for the optimisation inner loop of neobotix/neo_mpc_planner2.
 * The entry points below are what `NeoMpcPlanner::computeVelocityCommands`
 * (src/NeoMpcPlanner.cpp:240-252) binds INSTEAD of the blocking ROS2 service call
 * to the Python node: one `neo_mpc_problem` carries exactly the fields of the
 * `neo_srvs2/srv/Optimizer` request built at cpp:240-246, one `neo_mpc_command`
 * the `output_vel` returned at cpp:250-252, `neo_mpc_params` the ROS parameters the
 * node declares at neo_mpc_planner2/mpc_optimization_server.py:49-75, and
 * `neo_mpc_state` the state the node keeps between calls (py:115-152).
 *
 * Conventions: plain C, no exceptions; every call returns NEO_MPC_OK (0) or a
 * negative NEO_MPC_ERR_* code and `neo_mpc_last_error()` describes the failure;
 * the library never retains caller pointers after a call returns; a handle is NOT
 * thread-safe (the plugin already serialises under `mutex_`, cpp:207), distinct
 * handles are independent.  There is no CPU fallback: `neo_mpc_create` fails when
 * no gfx950 device is visible.
 */
#ifndef NEO_MPC_H_
#define NEO_MPC_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NEO_MPC_ABI_VERSION 2
/* ABI 2 (round 4): neo_mpc_behaviour_version(); NEO_MPC_COMPAT_REFERENCE_START; unknown compat bits and the forced
 * directions without a wall model at a heavy costmap weight are refused; neo_mpc_problem.skip (was reserved[0]) and
 * NEO_MPC_FLAG_SKIPPED; neo_mpc_carrot.status 3.  No record changed its size or the offset of a field that existed.
 * Round 5 added entry points only (neo_mpc_effective_method, neo_mpc_balance_dispatch_device) and gave the last 28 bytes
 * of neo_mpc_state, reserved until then, a meaning (has_prev_u0, prev_u0): still ABI 2.
 *
 * Behaviour history (iterates and iteration counts differ between versions, results stay inside the parity protocol of
 * DESIGN.md section 1; neo_mpc_behaviour_version() returns the number of the build that answers):
 *   1  AUTO = dense Newton up to 8 control steps, L-BFGS beyond.
 *   2  AUTO = dense Newton at control_steps 3, the stage-wise (Riccati) direction everywhere else -- and at 3 when
 *      w_costmap > w_trans / 4; stop thresholds beyond 3 control steps scaled with (3 / control_steps)^2.
 *   3  hop candidates in the stage-wise direction; blocked-run stop rule in the dense direction; searches on free space
 *      start from the better of the warm start and the warm start un-shifted; page-locked host batches are worked on in
 *      place (neo_mpc_solve_batch); neo_mpc_pin_host_memory, neo_mpc_set_host_path; neo_mpc_solve_batch_begin / _wait.
 *   4  the stage-wise direction carries the second-order terms of the rollout step (exact Hessian) behind an iteration
 *      won by a Newton step; its window rules judge runs of blocked iterations only; gain thresholds relative to the
 *      u-dependent part of the objective; blocks next to the control norm's kink take their proximal step on their face
 *      and sit every other Newton candidate out; a block sliding along a box bound stops at the disc corner; the dense
 *      direction tries a hop to a cheaper costmap cell where its search is about to end; state records are written back
 *      field by field (old_goal only when it changed).  Blocks in a corner of the feasible set that a Newton step sends
 *      outward are pinned and the direction is computed once more; the closing-in stop rule of the dense / L-BFGS
 *      directions waits for two blocked iterations; a step below opt_tolerance ends the search only if it won its iteration.
 *   5  (round 5) cell scan in place of the dense direction's exit hop, in both second-order directions: a search that has
 *      ended looks at the costmap cells around every stage (up to 3 cells away, inside the reach of a feasible rollout),
 *      evaluates the cheapest ones as candidates and is taken up again once when that gained more than opt_tolerance.
 *      The warm start is un-shifted with the previous solve's own first block (neo_mpc_state.prev_u0, was reserved); on a
 *      costmap the un-shifted point is a candidate of the first iteration instead of a starting point.  Closing-in stop
 *      rule, in free space: three real gains and a geometric estimate of what is left below the stall threshold.  A
 *      pinned LBFGS / NEWTON direction that a neo_mpc_set_params call takes across w_costmap = w_trans / 4 runs the
 *      stage-wise direction from there on (neo_mpc_effective_method) instead of failing the reconfigure; neo_mpc_create
 *      still refuses the combination.  neo_mpc_problem.skip: host batches with values other than 0 / 1 are refused.
 *   6  (round 6) AUTO at control_steps 3 (below w_costmap = w_trans / 4): DIRECTION BY NEIGHBOURHOOD.  An instance with a
 *      lethal cell (raw 254, or the outside of the map) among the cells of its reach tile -- NEO_MPC_FLAG_WALL_IN_REACH says
 *      so in its command -- is solved by the stage-wise direction (wall model, hop candidates; with the control_steps-3
 *      stop rules), every other instance by the dense direction as before; one launch, one kernel (k_solve_routed).  The
 *      dense direction has no wall model: every objective miss of random-parameter runs against the reference at
 *      control_steps 3 was a dense search hemmed in by lethal cells.  method = NEO_MPC_METHOD_NEWTON is the dense direction
 *      for every instance (version 5's AUTO at control_steps 3).
 * method = NEO_MPC_METHOD_NEWTON / _LBFGS / _RICCATI pins a direction. */
#define NEO_MPC_BEHAVIOUR_VERSION 6

/* return codes */
#define NEO_MPC_OK 0
#define NEO_MPC_ERR_INVALID_ARGUMENT (-1)
#define NEO_MPC_ERR_NO_DEVICE (-2)
#define NEO_MPC_ERR_DEVICE (-3)      /* a HIP runtime call failed */
#define NEO_MPC_ERR_NO_COSTMAP (-4)  /* solve before set_costmap */
#define NEO_MPC_ERR_UNSUPPORTED (-5)

/* per-instance solver status (neo_mpc_command.status); 0 <=> SciPy's `x.success` (py:397) */
#define NEO_MPC_STATUS_CONVERGED 0
#define NEO_MPC_STATUS_MAX_ITER 1

/* neo_mpc_command.flags */
#define NEO_MPC_FLAG_RESET 1   /* new goal: warm start / last_control / waiting_time reset (py:358-361) */
#define NEO_MPC_FLAG_STOPPED 2 /* zero twist because of the collision latch (py:374-377) */
#define NEO_MPC_FLAG_SKIPPED 4 /* neo_mpc_problem.skip was set: no request was made for this robot this tick (cpp:234-236).
                                  Its state record and warm start have not been touched; the command is zero twist with this
                                  flag alone, its rows of solution / predicted_path / velocities are zero */
#define NEO_MPC_FLAG_WALL_IN_REACH 8 /* a lethal cell (raw 254) -- or the outside of the map -- lies within the cells a feasible
                                  rollout of this robot can reach (the solver's reach tile; always set when the reach is too
                                  long for a tile).  Under NEO_MPC_METHOD_AUTO at control_steps 3 such an instance was solved
                                  by the stage-wise direction, every other one by the dense direction */

/* neo_mpc_params.compat_flags */
#define NEO_MPC_COMPAT_ODOM_YAW_GOAL_W 1 /* py:213: odom_yaw takes quaternion w from the goal pose */
#define NEO_MPC_COMPAT_REFERENCE_START 2 /* every search starts at the reference's starting point: the warm start as
                                            py:397-400 / 198-202 left it (projected onto the feasible set, like SciPy clips
                                            x0).  Without it (default) a search on free space starts from the better of that
                                            and the same sequence with the whole-step shift undone -- fewer iterations per
                                            warm tick, same objective */
#define NEO_MPC_COMPAT_ALL (NEO_MPC_COMPAT_ODOM_YAW_GOAL_W | NEO_MPC_COMPAT_REFERENCE_START) /* any other bit: INVALID_ARGUMENT */

/* neo_mpc_params.method */
#define NEO_MPC_METHOD_AUTO 0   /* control_steps == 3: by neighbourhood -- dense Newton (the register-resident 9 x 9
                                   system) for an instance with no lethal cell in reach, stage-wise (Riccati) Newton
                                   for one next to a wall (NEO_MPC_FLAG_WALL_IN_REACH), in one launch; stage-wise
                                   Newton at every other control_steps -- and for every instance at 3 when
                                   w_costmap > w_trans / 4 (cost steps become walls: the wall model is part
                                   of the stage-wise direction) */
#define NEO_MPC_METHOD_LBFGS 1  /* projected L-BFGS, any control_steps */
#define NEO_MPC_METHOD_NEWTON 2 /* projected Newton (finite-difference Hessian of the analytic
                                   gradient, one column per lane); control_steps <= 8 */
/* (LBFGS and NEWTON have no wall model for costmap steps: with w_costmap > w_trans / 4 -- where AUTO hands every
 * control_steps to the stage-wise direction -- they end above the reference's SLSQP on a few percent of the costmap cases
 * (G8 "turn", G9), so neo_mpc_create refuses that combination with NEO_MPC_ERR_UNSUPPORTED; a LIVE handle reconfigured
 * across the threshold (neo_mpc_set_params = cb_params, which cannot fail in the reference) keeps working: it runs the
 * stage-wise direction while the weights stay there -- neo_mpc_get_params still returns the pinned method,
 * neo_mpc_effective_method what runs) */
#define NEO_MPC_METHOD_RICCATI 3 /* projected Gauss-Newton, solved stage by stage (Riccati recursion over the
                                   rollout chain, 3x3 blocks, float32): any control_steps, O(control_steps)
                                   per iteration; beyond 8 control steps with adaptive Levenberg-Marquardt
                                   damping; in free space (no costmap term under the rollout) the full
                                   step is tried on its own before the 64-candidate search */
#define NEO_MPC_NEWTON_MAX_CONTROL_STEPS 8

#define NEO_MPC_MAX_CONTROL_STEPS 64
#define NEO_MPC_MAX_FOOTPRINT_POINTS 16
#define NEO_MPC_MAX_LBFGS_MEMORY 8

/* ROS parameters of the reference node, same names (py:49-75; README.md:53-84), then
 * this build's solver options. */
typedef struct neo_mpc_params {
  double acc_x_limit, acc_y_limit, acc_theta_limit;          /* post-clamp py:385-391 */
  double min_vel_x, min_vel_y, min_vel_trans, min_vel_theta; /* box bounds py:127-133 (min_vel_trans unused) */
  double max_vel_x, max_vel_y, max_vel_trans, max_vel_theta; /* max_vel_trans: disc constraint py:157-158 */
  double w_trans, w_orient, w_control, w_terminal;           /* py:252-253, 268 */
  double w_costmap, w_footprint;                             /* py:260, 263 */
  double waiting_time;                                       /* py:70 (initial value only, see py:361, 380) */
  double low_pass_gain;                                      /* py:367 */
  double opt_tolerance;                                      /* py:364 (SLSQP ftol) */
  double prediction_horizon;                                 /* py:137 */
  int32_t control_steps;                                     /* py:75 */
  /* --- build-specific --- */
  int32_t max_iterations; /* <=0: 100, SciPy SLSQP's maxiter */
  int32_t lbfgs_memory;   /* <=0: 4 */
  int32_t compat_flags;   /* NEO_MPC_COMPAT_*; neo_mpc_default_params sets NEO_MPC_COMPAT_ODOM_YAW_GOAL_W (the reference's
                             objective); bits outside NEO_MPC_COMPAT_ALL are refused */
  double step_tolerance;  /* stop when max|du| < this; <=0: 1e-3 * opt_tolerance, and (Newton) a full step
                             shorter than opt_tolerance -- SLSQP's step test -- is taken as the last one */
  double cost_tolerance;  /* an iteration is "stalled" when it lowers the objective by less than
                             cost_tolerance * max(1, |f|) (<=0: 3e-6 * opt_tolerance with L-BFGS, 3e-4 *
                             opt_tolerance with Newton) or moves less
                             than stall_step; 5 stalled iterations in a row end the search */
  double kink_radius;     /* blocks with |u_i - v_cur| below this are moved by the proximal step of
                             the control norm and kept out of the L-BFGS model; <=0: 3e-3 */
  double stall_step;      /* <=0: 0.3 * opt_tolerance */
  int32_t method;         /* NEO_MPC_METHOD_*: search direction of lanes 32-63 */
  int32_t reserved_i;
  double window_tolerance; /* the search also ends when three iterations in a row lower the objective by
                              less than window_tolerance * max(1, |f|) together (SLSQP ends on ONE
                              iteration gaining less than opt_tolerance); 0: 3e-3 * opt_tolerance with
                              Newton, off with L-BFGS (whose normal progress is that slow); <0: off */
} neo_mpc_params;

/* One Optimizer.srv request (cpp:240-246).  256 bytes. */
typedef struct neo_mpc_problem {
  double cur_xy[2];        /* current_pose.pose.position.{x,y}, costmap global frame (cpp:244) */
  double cur_q[4];         /* current_pose.pose.orientation x,y,z,w */
  double carrot_xy[2];     /* carrot_pose.pose.position.{x,y}, base frame (cpp:242) */
  double carrot_q[4];      /* carrot_pose.pose.orientation x,y,z,w */
  double goal_xyz[3];      /* goal_pose.position (cpp:243) */
  double goal_q[4];        /* goal_pose.orientation x,y,z,w */
  double cur_vel[3];       /* current_vel linear.x, linear.y, angular.z (cpp:241) */
  double control_interval; /* 1/controller_frequency (cpp:246) */
  double delta_t;          /* wall-clock seconds since the previous call (py:369-371) */
  double footprint_cost;   /* normalised getFootprintCost(published footprint) (py:262, 343); used
                              when the batch carries no polygons */
  int32_t map_index;       /* which costmap of a pool this instance lives in (neo_mpc_set_costmap_pool);
                              ignored with a single costmap */
  int32_t switch_opt;      /* request.switch_opt = closer_to_goal (cpp:245); the reference stores it (py:354)
                              and never reads it -- carried so that the record is the request field for field */
  int32_t skip;            /* 1: this robot makes NO request this tick -- the plugin threw before its service call
                              (footprint cost 255, cpp:234-236; neo_mpc_select_carrots sets it with every non-zero carrot
                              status): the solver leaves the robot's state and warm start alone (NEO_MPC_FLAG_SKIPPED).
                              Must be 0 or 1: the host entry points refuse anything else (NEO_MPC_ERR_INVALID_ARGUMENT;
                              the field was reserved[0] in ABI 1), the device entry points act on 1 alone */
  int32_t reserved_i;      /* reserved fields MUST be zero */
  double reserved[5];      /* (never read by the device: a request is 216 bytes on the wire) */
} neo_mpc_problem;

/* State the reference node keeps between requests (py:115-152).  128 bytes.  The warm
 * start (`initial_guess`, py:136) is a separate double[3*control_steps] row per instance. */
typedef struct neo_mpc_state {
  double last_control[3];      /* py:117 */
  double old_goal[7];          /* py:146, 402: position xyz + orientation xyzw */
  double waiting_time;         /* py:103, 361, 378-382 */
  int32_t has_old_goal;        /* 0 before the first call: py:146 compares PoseStamped with Pose */
  int32_t collision;           /* py:148 latch */
  int32_t collision_footprint; /* py:149 */
  int32_t has_prev_u0;         /* (was reserved_i) 1: prev_u0 holds ... */
  double prev_u0[3];           /* (was reserved[3]) ... the previous solve's first control block as the solver left it, before
                                  the low-pass of py:366-367 -- the node has no such attribute: it is the build's own hint
                                  (behaviour version 5), written by every solve / postprocess call and read by the next
                                  solve, which un-shifts the warm start with it.  It never changes
                                  what a result means: the point only competes by objective value; zeros (a fresh or an
                                  ABI-1 caller's record) and garbage are harmless */
} neo_mpc_state;

/* Optimizer.srv response (`output_vel.twist`, py:375-377, 389-391) + diagnostics.  48 bytes. */
typedef struct neo_mpc_command {
  double vel[3];       /* linear.x, linear.y, angular.z */
  double cost;         /* objective at the raw solver output (SciPy `x.fun`) */
  int32_t status;      /* NEO_MPC_STATUS_* */
  int32_t iterations;  /* `x.nit` */
  int32_t evaluations; /* objective evaluations per lane */
  int32_t flags;       /* NEO_MPC_FLAG_* */
} neo_mpc_command;

/* One batch of independent instances.  All pointers are host pointers for
 * neo_mpc_solve_batch / neo_mpc_postprocess_batch and device pointers for the
 * *_device variants.  Optional members may be NULL. */
typedef struct neo_mpc_batch {
  size_t count;
  const neo_mpc_problem* problems; /* [count] */
  neo_mpc_state* states;           /* [count] in/out */
  double* warm_start;              /* [count][3*control_steps] in/out (py:136, 397-400) */
  neo_mpc_command* commands;       /* [count] out */
  double* solution;                /* optional [count][3*control_steps]: raw solver output `x.x`
                                      (out for solve, IN for postprocess) */
  double* predicted_path;          /* optional out [count][control_steps][3]: X, Y, yaw of the
                                      `local_plan` rollout (py:293-306) */
  const double* footprints;        /* optional [count][footprint_points][2]: published footprint
                                      polygon, global frame (py:140-144) */
  uint32_t footprint_points;       /* 0: use problems[i].footprint_cost */
  uint32_t reserved;
  double* velocities;              /* optional out [count][3]: packed copy of commands[i].vel (the
                                      buffer a multi-GPU caller hands to the all-gather) */
} neo_mpc_batch;

/* ---- next row of the path: the carrot (look-ahead) selection that feeds the solver ---------- */

/* `<plugin>.lookahead_dist_*` (cpp:311-322) and the plan cut-off of cpp:78-80. */
typedef struct neo_mpc_lookahead_params {
  double lookahead_dist_min, lookahead_dist_max, lookahead_dist_close_to_goal;
  double max_transform_dist; /* max(size_x, size_y) * resolution / 2 (cpp:79-80) */
} neo_mpc_lookahead_params;

/* Result of transformGlobalPlan + getLookAheadDistance + getLookAheadPoint + the slow_down_
 * update (cpp:66-135, 157-189, 221-232) for one robot.  80 bytes. */
typedef struct neo_mpc_carrot {
  double xy[2];           /* carrot_pose.pose.position, base frame (cpp:214) */
  double q[4];            /* carrot_pose.pose.orientation x,y,z,w, base frame */
  double lookahead_dist;  /* cpp:211 */
  uint32_t begin, end;    /* plan poses [begin, end) were kept (cpp:83-104); the caller erases
                             [0, begin) like cpp:126 */
  int32_t closer_to_goal; /* cpp:95-100 (-> request.switch_opt, cpp:245) */
  int32_t slow_down;      /* slow_down_ after cpp:221-232 */
  int32_t status;         /* 0 ok; 1 plan with zero length (cpp:69-71); 2 nothing left (cpp:130-132); 3 the robot's
                             footprint cost is 255 (cpp:234-236: the plugin throws, no optimizer request is made --
                             carrot and look-ahead distance are filled in as for status 0, slow_down is updated like
                             cpp:221-232 did before the throw, problems[i].skip is set) */
  int32_t reserved;
} neo_mpc_carrot;

/* Ragged batch of global plans.  Poses are planar (x, y, yaw) in the plan frame; `robot_poses`
 * is the robot pose already expressed in that frame (transformPose, cpp:74-77). */
typedef struct neo_mpc_plan_batch {
  size_t count;
  const double* plan_poses;      /* [plan_offsets[count]][3] */
  const uint32_t* plan_offsets;  /* [count + 1] */
  const double* robot_poses;     /* [count][3] */
  const double* footprint_costs; /* [count] footprintCostAtPose on nav2's 0..255 scale (cpp:218) */
  int32_t* slow_down;            /* [count] in/out: slow_down_ (h:162, initially 1) */
  neo_mpc_carrot* carrots;       /* [count] out */
  neo_mpc_problem* problems;     /* optional [count]: carrot_xy / carrot_q are written into the
                                    requests the solver will consume (cpp:242) */
} neo_mpc_plan_batch;

typedef struct neo_mpc_handle neo_mpc_handle;

/* library / ABI */
int neo_mpc_abi_version(void);
int neo_mpc_behaviour_version(void);   /* NEO_MPC_BEHAVIOUR_VERSION of the library that answers */
const char* neo_mpc_last_error(void);
int neo_mpc_last_error_code(void);     /* the NEO_MPC_ERR_* of the calling thread's last failure (neo_mpc_create returns
                                          NULL and no code) */

/* Fills the defaults the reference node declares (py:49-75) and this build's solver options. */
int neo_mpc_default_params(neo_mpc_params* params);

/* Replaces `MpcOptimizationServer.__init__` (py:45-152) + the service client creation at
 * cpp:308.  `device` is the HIP device ordinal.  NULL on failure.
 * The A/B switches of the measurement tools are environment variables READ HERE, ONCE (never on the solve path; a tool
 * that flips one re-creates its handle): NEO_MPC_SOLVE_WAVES=2|3|4, NEO_MPC_NO_TAME_SPECIALISATION, NEO_MPC_DYNAMIC_LDS (kernel
 * variant), NEO_MPC_NO_CHUNKS (large staged host batches in one piece), NEO_MPC_HOST_PATH=staged|zerocopy|zerocopy_out (what
 * NEO_MPC_HOST_PATH_AUTO means).  None changes a result beyond rounding. */
neo_mpc_handle* neo_mpc_create(const neo_mpc_params* params, int device);
void neo_mpc_destroy(neo_mpc_handle* handle);

/* Dynamic reconfigure (`cb_params`, py:405-439).  Unlike the reference every field takes effect. */
int neo_mpc_set_params(neo_mpc_handle* handle, const neo_mpc_params* params);
int neo_mpc_get_params(const neo_mpc_handle* handle, neo_mpc_params* params);
/* (Cloning a live handle -- another GPU of a fleet server: neo_mpc_get_params returns what the caller SET, which may be a
 * pinned LBFGS / NEWTON that a later reconfigure took across w_costmap = w_trans / 4 -- a combination neo_mpc_create refuses.
 * Create the sibling with `method` = neo_mpc_effective_method(handle): that is the direction the live handle runs.) */
/* The direction the handle's solves run: NEO_MPC_METHOD_LBFGS / _NEWTON / _RICCATI (never AUTO) -- what AUTO resolved to,
 * or the stage-wise direction in place of a pinned LBFGS / NEWTON above w_costmap = w_trans / 4 (see NEO_MPC_METHOD_*);
 * < 0 on a null handle.  AUTO at control_steps 3 below that threshold decides per instance (behaviour 6): NEWTON is the
 * answer -- the direction of the instances with no wall in reach; those with NEO_MPC_FLAG_WALL_IN_REACH in their command ran
 * the stage-wise one. */
int neo_mpc_effective_method(const neo_mpc_handle* handle);

/* Replaces the node's `Costmap2d(self)` subscription (py:118): raw nav2 costs, row-major
 * cells[my*size_x + mx] (e.g. `costmap_->getCharMap()` in the plugin).  The data is copied before the
 * call returns; the device-side ingest is left in flight, ordered in front of every later call on
 * this handle (the *_device entry points wait for it on the caller's stream).
 * Stream ordering, all set_costmap* variants: the ingest records an event on the stream it ran on and every
 * solve / postprocess / objective entry point makes its own stream wait for it; every solve records an event
 * on its stream and the next ingest waits for it before it rewrites the device map -- a solve never sees a
 * half-written map whatever streams the caller mixes.  (The caller's own buffers -- `d_cells`, `d_origins`,
 * the batch arrays -- stay the caller's to order.)  The two small tables every wave reads as it starts -- the pool's origins
 * (neo_mpc_set_costmap_pool) and the per-step costmap terms (neo_mpc_set_params) -- are rewritten only after every launch
 * that may still read them has ended (a host-side wait): both calls are safe between neo_mpc_solve_batch_begin and _wait. */
int neo_mpc_set_costmap(neo_mpc_handle* handle, const uint8_t* cells, uint32_t size_x,
                        uint32_t size_y, double resolution, double origin_x, double origin_y);
/* Same, `d_cells` already in device memory; ingested on `stream` (hipStream_t, may be NULL). */
int neo_mpc_set_costmap_device(neo_mpc_handle* handle, const uint8_t* d_cells, uint32_t size_x,
                               uint32_t size_y, double resolution, double origin_x,
                               double origin_y, void* stream);

/* Fleet variant of neo_mpc_set_costmap: `count` costmaps of one size and resolution -- nav2's rolling
 * local costmaps, one per robot or per group of robots -- stored back to back (map k at
 * cells + k*size_x*size_y), origins[2k], origins[2k+1] = origin of map k.  Every instance reads the map
 * its `neo_mpc_problem.map_index` names.  Replaces whatever costmap(s) the handle held. */
#define NEO_MPC_MAX_POOL_MAPS 65535u /* one ingest launch: the map index is the grid's y coordinate */
int neo_mpc_set_costmap_pool(neo_mpc_handle* handle, const uint8_t* cells, uint32_t count, uint32_t size_x,
                             uint32_t size_y, double resolution, const double* origins);
/* Same with `d_cells` and `d_origins` in device memory; the ingest runs on `stream`.  `d_origins` is read
 * by every later solve: it must stay valid (and may be rewritten by the caller between ticks). */
int neo_mpc_set_costmap_pool_device(neo_mpc_handle* handle, const uint8_t* d_cells, uint32_t count,
                                    uint32_t size_x, uint32_t size_y, double resolution,
                                    const double* d_origins, void* stream);

/* Replaces `client->async_send_request(request); result.get()` (cpp:248-250), i.e. the whole of
 * `MpcOptimizationServer.optimizer` (py:349-403), for `count` independent instances.  Synchronous.
 * Pageable host arrays are staged through device memory (copies queued around the kernel, one wait; batches of
 * 65 536 instances or more in four pieces on two streams, so that copies and kernels overlap).  When EVERY
 * array of the batch is page-locked (hipHostMalloc, hipHostRegister, neo_mpc_pin_host_memory below, torch
 * pin_memory) nothing is copied: the kernel reads the records from the caller's arrays and writes the results into
 * them over PCIe while other instances compute -- one launch and one wait per call. */
int neo_mpc_solve_batch(neo_mpc_handle* handle, const neo_mpc_batch* batch);
/* The same call in its two halves -- `client->async_send_request(request)` and `result.get()` (cpp:248-250) -- for
 * callers that keep more than one batch moving (a server of several fleets; double-buffered ticks): `begin` enqueues
 * the batch on a stream of its own and returns a ticket, `wait` blocks until that batch's results are in its arrays.
 * Every array of the batch must be page-locked (it is worked on in place, see above; NEO_MPC_ERR_UNSUPPORTED
 * otherwise) and must not be touched between the two calls.  Up to NEO_MPC_MAX_BATCHES_IN_FLIGHT tickets at a time;
 * ticket 0 (an empty batch) needs no wait.  neo_mpc_set_costmap between begin and wait is ordered behind the batches in
 * flight.  Not thread-safe per handle, like every other call. */
#define NEO_MPC_MAX_BATCHES_IN_FLIGHT 4
int neo_mpc_solve_batch_begin(neo_mpc_handle* handle, const neo_mpc_batch* batch, uint32_t* ticket);
int neo_mpc_solve_batch_wait(neo_mpc_handle* handle, uint32_t ticket);
/* Page-locks `bytes` of host memory at `ptr` for the device (hipHostRegister) / releases it: lets a caller built
 * without HIP headers -- the nav2 plugin is plain g++ -- keep its request arena where neo_mpc_solve_batch can work on
 * it in place.  The caller unpins before it frees the memory. */
int neo_mpc_pin_host_memory(void* ptr, size_t bytes);
int neo_mpc_unpin_host_memory(void* ptr);
/* How neo_mpc_solve_batch moves a batch whose arrays are all page-locked (pageable arrays are always staged). */
#define NEO_MPC_HOST_PATH_AUTO 0          /* = ZEROCOPY (NEO_MPC_HOST_PATH=staged|zerocopy|zerocopy_out in the environment
                                             of neo_mpc_create overrides what AUTO means, for A/B runs) */
#define NEO_MPC_HOST_PATH_STAGED 1        /* copies into device staging and back (DMA), as for pageable arrays */
#define NEO_MPC_HOST_PATH_ZEROCOPY 2      /* the kernel reads and writes the caller's arrays in place */
#define NEO_MPC_HOST_PATH_ZEROCOPY_OUT 3  /* inputs copied up by DMA, results written in place by the kernel */
int neo_mpc_set_host_path(neo_mpc_handle* handle, int mode);
/* Same with every pointer in device memory; enqueued on `stream`, returns without waiting. */
int neo_mpc_solve_batch_device(neo_mpc_handle* handle, const neo_mpc_batch* batch, void* stream);

/* Balanced dispatch for a fleet's NEXT tick (round 5; optional, changes no result).  A launch of up to 4096 instances is one
 * residency round on an MI355X -- workgroups w, w + 1024, w + 2048, w + 3072 share a SIMD -- and ends with the SIMD whose
 * four searches need the most iterations.  Robots keep their habits from tick to tick, so the iteration counts of the
 * previous tick (`d_previous_commands[i].iterations`, device memory, `count` records) predict the next tick's load: the
 * library sorts the instances by them (by their exponential average over successive calls of the same count, decay 1/2),
 * deals them over the SIMDs longest first, and the following
 * neo_mpc_solve_batch_device[_timed] calls OF THE SAME COUNT on this handle solve instance order[w] in workgroup w (every
 * array of the batch stays in the caller's order; every instance's result is bit for bit what it is without).  The order
 * is built by a small kernel on `stream` -- enqueue the next solve behind it, i.e. on the same stream or behind an event.
 * Refresh it every few ticks; counts that are not a multiple of 1024, or beyond 4096, get launch order.
 * d_previous_commands == NULL: back to launch order. */
int neo_mpc_balance_dispatch_device(neo_mpc_handle* handle, const neo_mpc_command* d_previous_commands, size_t count,
                                    void* stream);
/* Same; `start_event` / `stop_event` (hipEvent_t, either may be NULL) are stamped with the start and the
 * end of the solve kernel by the dispatch itself (hipExtLaunchKernel): a measurement harness gets K1's
 * duration without putting event-record packets between consecutive launches. */
int neo_mpc_solve_batch_device_timed(neo_mpc_handle* handle, const neo_mpc_batch* batch, void* stream,
                                     void* start_event, void* stop_event);

/* Only the part of `optimizer` after the solve (py:365-403): low-pass, collision check, stop
 * latch, acceleration clamp, warm-start shift, with `batch->solution` supplying `x.x` and
 * `success[i]` supplying `x.success` (NULL: all true).  Host pointers. */
int neo_mpc_postprocess_batch(neo_mpc_handle* handle, const neo_mpc_batch* batch,
                              const int32_t* success);

/* Test hook: the total gradient the solve kernel works with at u[count][3*control_steps] (projected onto the
 * feasible set first, like x0): analytic adjoint gradient of the tracking + terminal cost plus the gradient of
 * the control norm, taken from inside the kernel variant the current parameters select.  What SciPy obtains
 * by forward differences of `objective` (py:204-269; _slsqp_py.py:381).  Host pointers. */
int neo_mpc_gradient_batch(neo_mpc_handle* handle, const neo_mpc_problem* problems, const double* u,
                           double* grad_out, size_t count);

/* Test hook: the search direction of lanes 32-63 (Newton / L-BFGS) in solver iteration `iteration` (0-based) of a
 * solve started from u (instances that stop earlier leave their row untouched).  Host pointers. */
int neo_mpc_direction_batch(neo_mpc_handle* handle, const neo_mpc_problem* problems, const double* u,
                            double* dir_out, size_t count, int iteration);

/* `MpcOptimizationServer.objective` (py:204-269) evaluated on the device for
 * u[count][3*control_steps] (not projected); `footprint_cost` from problems[i].  Host pointers. */
int neo_mpc_objective_batch(neo_mpc_handle* handle, const neo_mpc_problem* problems,
                            const double* u, double* cost_out, size_t count);

/* Carrot selection for `count` robots (host pointers / device pointers + stream): replaces
 * transformGlobalPlan's pruning, getLookAheadDistance, getLookAheadPoint and the slow_down_ state
 * machine (cpp:83-104, 157-189, 221-232) with the plan->base transform taken as the planar rigid
 * transform given by the robot pose. */
int neo_mpc_select_carrots(neo_mpc_handle* handle, const neo_mpc_lookahead_params* params,
                           const neo_mpc_plan_batch* batch);
int neo_mpc_select_carrots_device(neo_mpc_handle* handle, const neo_mpc_lookahead_params* params,
                                  const neo_mpc_plan_batch* batch, void* stream);

/* ---- multi-GPU fleets: the one exchange step ------------------------------------------------
 * Instances of one tick shard embarrassingly over the GPUs of a node (one handle per GPU, costmap and parameters
 * replicated, per-instance state resident on its GPU); what `computeVelocityCommands` returns for every robot
 * (cpp:251-254) is collected with ONE all-gather of the packed `neo_mpc_batch.velocities` over RCCL (xGMI).
 * `comm` is an ncclComm_t of the caller's, or one made by neo_mpc_comm_init_all (a single process driving all
 * devices: bracket the per-device calls with neo_mpc_group_start / neo_mpc_group_end).  RCCL is bound at run time;
 * without librccl.so these return NEO_MPC_ERR_UNSUPPORTED and everything else keeps working.  No torch types. */
int neo_mpc_rccl_available(void);
int neo_mpc_comm_init_all(int ndev, const int* devices, void** comms_out);   /* ncclCommInitAll */
int neo_mpc_comm_destroy(void* comm);
int neo_mpc_group_start(void);
int neo_mpc_group_end(void);
/* d_local[count][3] of this rank -> d_all[world][count][3] on every rank, enqueued on `stream` (hipStream_t). */
int neo_mpc_allgather_velocities(const double* d_local, double* d_all, size_t count, void* comm, void* stream);
/* One-off: rank `root`'s raw costmap cells to every rank (in place), then neo_mpc_set_costmap_device per rank. */
int neo_mpc_broadcast_costmap(uint8_t* d_cells, size_t bytes, int root, void* comm, void* stream);

/* Bytes of LDS and costmap reach (cells) the solve kernel uses with the current params/map. */
int neo_mpc_kernel_info(const neo_mpc_handle* handle, uint32_t* lds_bytes, uint32_t* reach_cells,
                        uint32_t* tile_in_lds);

#ifdef __cplusplus
}
#endif
#endif /* NEO_MPC_H_ */
