/*
 * smpc.h — C ABI of the MI355X batched social-MPC solver.
 *
 * This is the drop-in boundary for ONE hot path of PIC4SeR/nav2_social_mpc_controller:
 * everything `Optimizer::optimize` does between `project_people(...)` returning and the
 * `TrajectoryMemory` store, i.e. reference `src/optimizer.cpp:197-446`:
 *   - problem assembly            (src/optimizer.cpp:241-379)   -> implied by smpc_params + smpc_scene_batch
 *   - ceres::Solve                (src/optimizer.cpp:381)       -> smpc_solve_batch
 *   - per-residual functors       (include/.../critics/<critic>_cost_function.hpp, update_state.hpp) -> smpc_eval_batch / inside solve
 *   - post-solve unpack + re-roll (src/optimizer.cpp:390-446)   -> smpc_result_batch.cmds / .path
 *
 * Plain C types only, caller-allocated buffers, no exceptions cross this boundary.
 * All floating point data are IEEE f64, costmaps are u8 (nav2 `Costmap2D::getCharMap()` layout,
 * row-major [size_y][size_x]).
 *
 * Notation: T = rollout steps (= optim_velocities.size() after pop_back,
 * src/optimizer.cpp:237), CH = min(control_horizon,T), bl = min(parameter_block_length,CH),
 * nb = (CH-1)/bl+1 parameter blocks, P = 2*nb unknowns [v_0, w_0, v_1, w_1, ...],
 * N = agents per step, M = residual count (8T or 5T per-step rows + CH/bl-1 feasibility rows).
 */
#ifndef SMPC_H_
#define SMPC_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SMPC_ABI_VERSION 4
#define SMPC_MAX_BLOCKS 10 /* nb <= 10  => P <= 20; every nb in 1..10 is instantiated */
#define SMPC_MAX_LM_ITERATIONS 100000 /* smpc_create refuses a larger max_iterations: a persistent wave must reach its exit */
#define SMPC_MAX_STEPS 63  /* T <= 63: one lane per pose of the rollout (T + 1 poses in a 64-lane wavefront) */
#define SMPC_MAX_AGENTS 64 /* N <= 64: one bit per agent in the per-step validity mask */

/* linear_solver_type: mirrors OptimizerParams::solver_types (optimizer.hpp:71-77). The value is validated like the
 * reference does (src/optimizer.cpp:31-45) and otherwise has NO effect on the device: every type takes the same step
 * solver — damped normal equations (J^T J + D^T D) delta = -J^T r by Cholesky (csrc/smpc_lm.hpp). For these problems
 * (P <= 20 unknowns, Jacobi-scaled) that is the LM step DENSE_QR / DENSE_SCHUR compute up to round-off; the CPU oracle
 * implements Householder QR and Schur elimination separately and tests compare against both. */
enum smpc_linear_solver {
  SMPC_DENSE_SCHUR = 0,
  SMPC_SPARSE_SCHUR = 1,
  SMPC_DENSE_NORMAL_CHOLESKY = 2,
  SMPC_DENSE_QR = 3,
  SMPC_SPARSE_NORMAL_CHOLESKY = 4
};

/* Per-scene termination, mirrors ceres::TerminationType as far as Optimizer::optimize observes it
 * (summary.IsSolutionUsable(), src/optimizer.cpp:384). */
enum smpc_status {
  SMPC_NOT_SOLVED = -1,    /* only with a device-side `order` that is not a permutation: the scene was never handed out */
  SMPC_CONVERGENCE = 0,    /* usable */
  SMPC_NO_CONVERGENCE = 1, /* usable (iteration cap) */
  SMPC_FAILURE = 2         /* NOT usable: non-finite initial evaluation or 5 consecutive invalid steps */
};

/* Reason detail for SMPC_CONVERGENCE / others (diagnostics only). */
enum smpc_reason {
  SMPC_REASON_NONE = 0,
  SMPC_REASON_GRADIENT_TOL = 1,
  SMPC_REASON_PARAMETER_TOL = 2,
  SMPC_REASON_FUNCTION_TOL = 3,
  SMPC_REASON_MIN_RADIUS = 4,
  SMPC_REASON_MAX_ITERATIONS = 5,
  SMPC_REASON_INVALID_STEPS = 6,
  SMPC_REASON_EVAL_FAILED = 7,
  SMPC_REASON_SHORT_PATH = 8 /* T_scene[b] < 1: "Path has less than 2 points, cannot optimize" (src/optimizer.cpp:158-162) */
};

/* API return codes (never exceptions). */
enum smpc_error {
  SMPC_OK = 0,
  SMPC_ERR_INVALID_ARG = -1,
  SMPC_ERR_UNSUPPORTED = -2, /* nb > SMPC_MAX_BLOCKS, T > SMPC_MAX_STEPS or N > SMPC_MAX_AGENTS */
  SMPC_ERR_DEVICE = -3,      /* HIP runtime error; see smpc_last_error() */
  SMPC_ERR_NO_DEVICE = -4
};

/* Mirrors OptimizerParams (optimizer.hpp:59-101) + the constants Optimizer::optimize hard-codes. */
typedef struct smpc_params {
  /* weights.* (src/optimizer.cpp:57-75) */
  double distance_w;
  double socialwork_w;
  double velocity_w;
  double angle_w; /* drives the second DistanceCost (path-align), src/optimizer.cpp:333-334 */
  double agent_angle_w;
  double proxemics_w;
  double velocity_feasibility_w;
  double obstacle_w;
  double goal_align_w;
  /* optimizer.* */
  int control_horizon;        /* raw parameter; clamped to T per batch */
  int parameter_block_length; /* raw parameter; clamped to CH */
  int max_iterations;         /* Solver::Options::max_num_iterations (src/optimizer.cpp:118) */
  int linear_solver_type;     /* enum smpc_linear_solver */
  double fn_tol;
  double gradient_tol;
  double param_tol;
  /* constants that are literals in the reference; exposed with the same defaults */
  double desired_linear_vel; /* 0.6, src/optimizer.cpp:238 */
  double v_min, v_max;       /* 0.0, 0.6, src/optimizer.cpp:375-376 */
  double w_min, w_max;       /* -1.4, 1.4, src/optimizer.cpp:377-378 */
  /* benchmark / diagnostics switches (0 = reference behaviour) */
  int fixed_iterations; /* !=0: disable the three tolerance tests, run exactly max_iterations LM iterations */
  int tol_needs_successful_step; /* 0: Ceres 2.0.x order; 1: Ceres >=2.1 (tolerance exits need one accepted step first) */
} smpc_params;

/* Fill with the reference's code defaults (src/optimizer.cpp:26-82) and hard-coded literals. */
void smpc_params_default(smpc_params* p);

/* One batch of independent scenes. All scenes share T, N, dt, costmap geometry.
 * Row layout of every scene: 8 rows per step when the scene has people, 5 when it has none (has_people[i] == 0),
 * plus the feasibility rows; output arrays of smpc_eval_batch are always strided by M = 8 T + n_feasibility (the
 * with-people size, whatever has_people says), rows a scene does not have are written as zeros. */
typedef struct smpc_scene_batch {
  int32_t B;         /* scenes */
  int32_t T;         /* rollout steps */
  int32_t N;         /* agents per step (reference: exactly 3, padded with t=-1) */
  int32_t on_device; /* 0: all pointers are host memory; 1: all pointers are device (HBM) memory */
  double dt;         /* time_step, already widened from float as the reference does (optimizer.hpp:170) */

  const double* pose0;       /* [B][3]  x, y, yaw of evolving_poses[0]                (src/optimizer.cpp:219-226) */
  const double* init_params; /* [B][P]  initial block values = optim_velocities[0..nb-1] (aliasing quirk, :254-261) */
  const double* path_pts;    /* [B][T+1][2] optim_positions; [T] is the final trajectorized point (:234-235,327) */
  const double* goal_yaw;    /* [B]     optim_headings.back().params[1]               (:298) */
  const double* people;      /* [B][T+1][6][N] people_proj: fields x,y,yaw,t,lv,av; t == -1 marks invalid (:193) */
  const uint8_t* has_people; /* [B]     people.people.size() != 0                     (:263) */

  const uint8_t* costmap;       /* [B or 1][size_y][size_x] */
  int32_t costmap_shared;       /* 1: one costmap for all scenes */
  int32_t size_x, size_y;       /* getSizeInCellsX/Y */
  const double* costmap_origin; /* [B][2] or [1][2] when shared: getOriginX/Y */
  double resolution;            /* getResolution */

  /* Optional (both or neither; same memory space): the people block in the form the sweep reads, as written by
   * smpc_stage_people_batch for the same pose0 / people / has_people. When NULL the library stages `people` itself
   * (one extra kernel per call). When given, `people` is not read and may be NULL. */
  const double* people_records; /* [B][N][T][4] px, py, lv cos(yaw), lv sin(yaw) of people_proj[t+1][a] */
  const double* people_aux;     /* [B][T][2]    bits of the valid-agent mask (t != -1), agent-angle target or SMPC_NO_TARGET */

  /* Optional scheduling hint of smpc_solve_batch: the order in which the persistent kernel's queue hands out the
   * scenes, a permutation of 0..B-1 (checked for host arrays; for device arrays a value outside 0..B-1 is skipped and a
   * repeated one is solved twice — the scene such an order leaves out is not solved: its status reads SMPC_NOT_SOLVED
   * and its other result rows are unspecified). Results do not depend on it (every scene is solved by its own lanes
   * from its own inputs); the duration of a launch that has the GPU to itself does: scenes that need many sweeps
   * should come first, e.g. sorted by the `evaluations` of the previous control period (a launch in index order ends
   * with a tail of late-started long scenes: 3.25 ms against 2.40 ms longest-first at the headline batch). NULL:
   * index order. */
  const int32_t* order; /* [B] */

  /* Optional: the rollout steps of each scene, 1 <= T_scene[b] <= T (checked for host arrays; device arrays are clamped
   * into that range by the kernel). The reference solves whatever horizon the tick produces — T = optim_velocities.size()
   * after the pop_back (src/optimizer.cpp:237), one less than the poses format_to_optimize kept (:492-497) — so a robot
   * whose trajectorized path ends before the batch's horizon (it is about to reach its goal) is solved with its own
   * T_b, CH_b = min(control_horizon, T_b), bl_b = min(parameter_block_length, CH_b) (:248-249) and the block, bound and
   * feasibility-row counts that follow from them. Every array keeps the batch's strides (T + 1 poses, P parameters): of
   * path_pts / people the first T_b + 1 entries of scene b are read (its final point is path_pts[b][T_b]), of
   * init_params the first P_b; results: params[P_b..P) = 0, cmds / path rows beyond T_b are written as zeros.
   * T_scene[b] < 1 (a path of fewer than two poses, for which Optimizer::optimize returns false, :158-162): status
   * SMPC_FAILURE with reason SMPC_REASON_SHORT_PATH. NULL: every scene has T steps. */
  const int32_t* T_scene; /* [B] */
} smpc_scene_batch;

#define SMPC_NO_TARGET 1e300 /* people_aux: AgentAngleCost is inactive at this step */

typedef struct smpc_result_batch {
  /* Any pointer may be NULL (that output is skipped). Same memory space as the scene batch. */
  double* params;       /* [B][P]       optimised block values */
  double* cmds;         /* [B][T+1][2]  saving_velocities (v, w)                   (src/optimizer.cpp:395-419) */
  double* path;         /* [B][T+1][3]  re-rolled x, y, yaw; pose0 omitted          (:420-446) */
  int32_t* status;      /* [B] enum smpc_status */
  int32_t* reason;      /* [B] enum smpc_reason */
  int32_t* iterations;  /* [B] LM iterations performed (successful + unsuccessful) */
  int32_t* evaluations; /* [B] residual+Jacobian sweeps performed */
  double* initial_cost; /* [B] */
  double* final_cost;   /* [B] */
} smpc_result_batch;

/* Outputs of the stand-alone residual/Jacobian sweep (kernel K1), used for parity checks and roofline runs. */
typedef struct smpc_eval_batch_out {
  double* residuals; /* [B][M]    reference residual order */
  double* jacobian;  /* [B][M][P] dense, unscaled */
  double* cost;      /* [B] 0.5*||r||^2 */
  double* gradient;  /* [B][P] J^T r */
  /* Row order of residuals / jacobian. 0: the reference's (step-major, SURVEY §8 a10: per step i the critics in the
   * order of src/optimizer.cpp:263-363, the feasibility row of block i behind step i). 1: critic-major — row of
   * critic c at step i at index c * T + i (c = 0..7 with people, 0..4 without, same critic order), the feasibility
   * rows behind them at rows_per_step * T + (block - 1): the same rows, laid out so that the T rows of one critic are
   * one contiguous block that the kernel writes as whole 128-byte lines. */
  int32_t row_order;
} smpc_eval_batch_out;

/* People projection that feeds the hot path: Optimizer::project_people + computeObstacle
 * (src/optimizer.cpp:554-728) with the Social Force Model of sfm.hpp, for B scenes at once. */
typedef struct smpc_projection_batch {
  int32_t B;
  int32_t T;         /* robot_path has T+1 states; T projection steps */
  int32_t N;         /* agents per scene (reference: 3 after people_to_status) */
  int32_t on_device; /* 0: host pointers, 1: device pointers (outputs follow) */
  float max_time;    /* naive_goal_time = trajectorizer.max_time (src/optimizer.cpp:558) */
  float time_step;   /* the float the reference passes down (optimizer.hpp:170) */
  const double* init_people; /* [B][N][6]    people_to_status output: x,y,yaw,t,lv,av; t == -1 marks invalid */
  const double* robot_path;  /* [B][T+1][6]  format_to_optimize output (optim_status) */
  const uint32_t* od_indexes; /* [B or 1][od_height][od_width] ObstacleDistance.indexes */
  int32_t od_shared;          /* 1: one distance grid (and origin) for all scenes */
  int32_t od_width, od_height;
  float od_resolution;
  const double* od_origin;    /* [B or 1][2] ObstacleDistance.info.origin.position.{x,y} */
} smpc_projection_batch;

/* per-scene outcome of smpc_project_people_batch: where the reference would throw std::runtime_error */
enum smpc_projection_error {
  SMPC_PROJ_OK = 0,
  SMPC_PROJ_CELL_OUT_OF_BOUNDS = 1,  /* src/optimizer.cpp:693-700 */
  SMPC_PROJ_INDEX_OUT_OF_BOUNDS = 2  /* :707-713 */
};

typedef struct smpc_handle smpc_handle;

/* Problem dimensions implied by (params, T, has_people): fills any non-NULL output. Returns smpc_error. */
int smpc_dims(const smpc_params* p, int T, int has_people, int* CH, int* bl, int* nb, int* P, int* M,
              int* n_bounded_blocks);

/* Create a solver bound to HIP device `device`. Fails (NULL, smpc_last_error()) when no device is present:
 * there is no CPU fallback behind this ABI. */
smpc_handle* smpc_create(const smpc_params* p, int device);
void smpc_destroy(smpc_handle* h);

/* Stream the kernels are enqueued on (a hipStream_t passed as void*); NULL = default stream. */
int smpc_set_stream(smpc_handle* h, void* hip_stream);

/* Solve every scene of the batch: replaces src/optimizer.cpp:241-446 for B scenes at once.
 * on_device=1: asynchronous on the handle's stream. on_device=0: stages through device memory and
 * returns after the results are back in host memory. */
int smpc_solve_batch(smpc_handle* h, const smpc_scene_batch* scenes, smpc_result_batch* out);

/* How many solve launches share the GPU at a time (this handle's and those of other handles on other streams, e.g.
 * the shards of a closed-loop batch: every shard's chain of small kernels must find free wave slots next to the other
 * shards' persistent solve kernels). The persistent grid of smpc_solve_batch is sized to 1/n of the wavefronts that
 * fit on the device; n = 1 (default): a launch sized for having the GPU to itself. Returns SMPC_ERR_INVALID_ARG for
 * n < 1. */
int smpc_set_solve_share(smpc_handle* h, int32_t n);

/* Lanes of a wavefront one scene takes in a solve launch of B scenes with T rollout steps and N agents on this handle:
 * 32 = two scenes per wave, 64 = one scene per wave (a shape with more than 31 steps or 32 agents always; a shape that
 * would fit two per wave while the batch is small enough for every scene to have a wave of its own anyway — the lanes
 * beyond the horizon then work as helper lanes in the agent loop, which shortens every sweep: the plugin's own B = 1
 * call, BASELINE configs[1]). A function of (B, T, N, the handle's solve share) alone, so a scene's result never depends
 * on timing; with four or more agents the two widths differ in the last bits of the sums over the agents. Returns a
 * negative smpc_error for a shape smpc_solve_batch would refuse. */
int smpc_solve_slot_width(const smpc_handle* h, int32_t B, int32_t T, int32_t N);

/* Evaluate residuals / Jacobian at `params` ([B][P], same memory space) — kernel K1 alone. */
int smpc_eval_batch(smpc_handle* h, const smpc_scene_batch* scenes, const double* params,
                    smpc_eval_batch_out* out);

/* Stage the people block of `scenes` (reads B, T, N, on_device, pose0, people, has_people) into the form the sweep
 * reads: records [B][N][T][4] and aux [B][T][2] (layouts: smpc_scene_batch.people_records / people_aux). Everything the
 * critics need from people_proj that does not depend on the optimised parameters is computed here, once per people
 * block: the agents' velocity vectors (social_work_cost_function.hpp:187-188), the validity test (:175) and the
 * steering target of AgentAngleCost (agent_angle_cost_function.hpp:130-190). */
int smpc_stage_people_batch(smpc_handle* h, const smpc_scene_batch* scenes, double* records, double* aux);

/* Roll the people forward with the Social Force Model: people_proj [B][T+1][6][N] (the layout smpc_scene_batch.people
 * expects; entry 0 = init_people, valid agents compacted to the front, the rest padded with t = -1 like the reference),
 * error [B] (enum smpc_projection_error; may be NULL). Returns SMPC_ERR_INVALID_ARG for an empty / malformed grid
 * (the reference throws, src/optimizer.cpp:676-687). */
int smpc_project_people_batch(smpc_handle* h, const smpc_projection_batch* in, double* people_proj, int32_t* error);

/* ---- warm start / input formatting for B scenes ------------------------------------------------
 * TrajectoryMemory (trajectory_memory.hpp:30-49; a process-wide singleton in the reference) as one caller-owned record
 * per scene. Yaws are stored as doubles, as tf2::getYaw reads the stored orientation back. */
typedef struct smpc_memory_batch {
  double* prev_path;  /* [B][T+1][3] previous_path.poses: x, y, yaw */
  double* prev_cmds;  /* [B][T+1][2] previous_cmds: linear.x, angular.z */
  int32_t* valid;     /* [B] 0 while previous_path.poses.size() == 0 (src/optimizer.cpp:177) */
  int32_t* length;    /* [B][2] previous_path.poses.size(), previous_cmds.size() of each record (as far as its T + 1 rows
                         hold them), kept up to date by smpc_format_to_optimize_batch and smpc_memory_store_batch: needed
                         when the scenes have horizons of their own (smpc_format_batch.n_poses), since format_to_optimize
                         blends pose i only while i < previous_path.poses.size() (:504). NULL: fixed horizon, every
                         record holds T + 1 of both. */
} smpc_memory_batch;

/* Optimizer::people_to_status (src/optimizer.cpp:454-482) for B scenes: people_msgs::Person position / velocity ->
 * AgentStatus rows (x, y, yaw = atan2(vy, vx), t = 0, lv = |v|, av = velocity.z), padded with invalid agents (t = -1) or
 * truncated to N agents (the reference hard-codes N = 3). has_people = people.people.size() != 0 (:263).
 * Optionally preceded by the field-of-view filter of SocialMPCController::computeVelocityCommands
 * (src/social_mpc_controller.cpp:196-214): a person is kept when Costmap2D::worldToMap accepts the
 * position and |shortest_angular_distance(robot yaw, bearing to the person)| < fov_angle, float arithmetic as there. */
typedef struct smpc_people_batch {
  int32_t B;
  int32_t Np;        /* row stride of `people`: the most persons any scene has (>= 1) */
  int32_t N;         /* agents per scene in the output */
  int32_t on_device; /* 0: host pointers, 1: device pointers (outputs follow) */
  const double* people;  /* [B][Np][5] position.x, position.y, velocity.x, velocity.y, velocity.z */
  const int32_t* count;  /* [B] persons of each scene */
  /* field-of-view filter; robot_pose == NULL: none (every person goes to people_to_status) */
  const double* robot_pose;      /* [B][3] x, y, tf2::getYaw(orientation) */
  double fov_angle;              /* SocialMPCController::fov_angle_ (default pi/4, :60) */
  const double* costmap_origin;  /* [B or 1][2] Costmap2D origin */
  int32_t costmap_shared;        /* 1: one origin for all scenes */
  int32_t size_x, size_y;        /* Costmap2D size in cells */
  double resolution;
} smpc_people_batch;

int smpc_people_to_status_batch(smpc_handle* h, const smpc_people_batch* in, double* init_people /* [B][N][6] */,
                                uint8_t* has_people /* [B]; may be NULL */);

/* Optimizer::format_to_optimize (src/optimizer.cpp:484-551) for B scenes whose incoming path has already been cut to
 * T + 1 poses (the cut to round(max_time / time_step) - 1 poses, :492-497, is a host-side length decision), followed by
 * what Optimizer::optimize derives from its result before building the problem (:197-261). Scenes with an empty memory
 * record first store the incoming path / cmds in it (:177-183) and then blend with that copy, like the reference. */
typedef struct smpc_format_batch {
  int32_t B;
  int32_t T;          /* T + 1 poses of every path are formatted */
  int32_t path_rows;  /* row stride of `path` / `cmds` in poses (0: T + 1) — e.g. max_steps + 1 when they come straight from
                         smpc_trajectorize_path_batch and T + 1 is the cut length */
  int32_t on_device;  /* 0: host pointers (memory record included), 1: device pointers; outputs follow */
  float time_step;
  float current_path_w; /* OptimizerParams::current_path_w / current_cmds_w (optimizer.hpp:93-94, floats) */
  float current_cmds_w;
  const double* path;  /* [B][path_rows][3] x, y, tf2::getYaw(orientation) of the trajectorizer path */
  const double* cmds;  /* [B][path_rows][2] trajectorizer commands (entries 0..T are read; T only to seed an empty memory) */
  const double* speed; /* [B][2] current robot twist linear.x, angular.z */
  smpc_memory_batch memory;
  /* Horizons per scene (optional): n_poses[b] = poses of the incoming path of scene b (smpc_trajectorize_out.n_poses;
   * its commands: one fewer), max_poses = (int)round(max_time / time_step) of the cut (:491-497): a path of more than
   * max_poses poses keeps max_poses - 1, any other path all of its poses (one of exactly max_poses poses is NOT cut: its
   * horizon is one step longer than that of the longer paths — size T for it: T = max_poses - 1). The scene's horizon
   * T_b = kept poses - 1 is returned in smpc_format_out.T_scene; rows a scene does not have are written as zeros.
   * n_poses == NULL: every path is taken to have T + 1 poses, all of them kept. */
  const int32_t* n_poses; /* [B] */
  int32_t max_poses;      /* 0: no cut */
} smpc_format_batch;

typedef struct smpc_format_out {
  double* robot_status; /* [B][T+1][6] optim_status: x, y, yaw, t, lv, av (input of smpc_project_people_batch) */
  double* pose0;        /* [B][3]      -> smpc_scene_batch.pose0 */
  double* init_params;  /* [B][P]      -> smpc_scene_batch.init_params (P from smpc_dims) */
  double* path_pts;     /* [B][T+1][2] -> smpc_scene_batch.path_pts */
  double* goal_yaw;     /* [B]         -> smpc_scene_batch.goal_yaw: heading of the scene's last kept pose */
  int32_t* T_scene;     /* [B]         -> smpc_scene_batch.T_scene (0: fewer than two poses); may be NULL */
} smpc_format_out;

int smpc_format_to_optimize_batch(smpc_handle* h, const smpc_format_batch* in, smpc_format_out* out);

/* The TrajectoryMemory store at the end of Optimizer::optimize (src/optimizer.cpp:448-449): scenes whose solve was
 * usable (status != SMPC_FAILURE; the reference returns before the store otherwise, :384-388) keep the optimised path
 * and commands for the next call. path / cmds / status are smpc_result_batch arrays. */
int smpc_memory_store_batch(smpc_handle* h, int32_t B, int32_t T, int32_t on_device, const int32_t* status,
                            const double* path, const double* cmds, smpc_memory_batch* memory,
                            const int32_t* T_scene /* [B] horizons of the solve (its T_scene), or NULL: T everywhere */);

/* ---- initial-guess generator -------------------------------------------------------------------
 * PathTrajectorizer::trajectorize (src/path_trajectorizer.cpp:120-288; motion model path_trajectorizer.hpp:106-135)
 * for B global plans: pure-pursuit simulation from the robot pose until the plan's last pose is within 0.2 m or
 * max_steps steps were taken. Parameters mirror PathTrajectorizer::configure (src/path_trajectorizer.cpp:53-84). */
enum smpc_trajectorize_error {
  SMPC_TRAJ_OK = 0,
  SMPC_TRAJ_SHORT_PLAN = 1,  /* fewer than 2 poses: trajectorize returns false (:123-127); n_poses = 0 */
  SMPC_TRAJ_NO_WAYPOINT = 2  /* every plan pose farther than 100 m and outside the look-ahead circle: the reference
                                indexes poses[-1] (:160-178); the simulation stops at the step it happens */
};

typedef struct smpc_trajectorize_batch {
  int32_t B;
  int32_t L;          /* row stride of `plan`: the longest plan, in poses */
  int32_t max_steps;  /* (int)round(max_time / time_step) (:84) */
  int32_t on_device;  /* 0: host pointers, 1: device pointers (outputs follow) */
  int32_t omnidirectional;
  double desired_linear_vel; /* defaults 0.4, 0.4, 1.0, 0.05 (:53-60) */
  double lookahead_dist;
  double max_angular_vel;
  double time_step;
  const double* plan;        /* [B][L][2] plan pose positions, in the frame of robot_pose */
  const int32_t* plan_len;   /* [B] poses of each plan */
  const double* robot_pose;  /* [B][3] x, y, tf2::getYaw(orientation) */
} smpc_trajectorize_batch;

typedef struct smpc_trajectorize_out {
  double* path;     /* [B][max_steps+1][3] x, y, yaw as tf2::getYaw reads the stored orientation; pose 0 = robot pose;
                       same layout as smpc_format_batch.path */
  double* cmds;     /* [B][max_steps+1][2] linear.x, angular.z of every step taken; same layout as smpc_format_batch.cmds */
  double* cmds_vy;  /* [B][max_steps+1] linear.y (non-zero for omnidirectional only); may be NULL */
  int32_t* n_poses; /* [B] poses written = steps taken + 1; rows beyond are zero */
  int32_t* error;   /* [B] enum smpc_trajectorize_error; may be NULL */
} smpc_trajectorize_out;

int smpc_trajectorize_path_batch(smpc_handle* h, const smpc_trajectorize_batch* in, smpc_trajectorize_out* out);

/* ---- the plan window of mpc::PathHandler::transformGlobalPlan (reference src/path_handler.cpp:39-108),
 * the step of computeVelocityCommands in front of the trajectorizer (src/social_mpc_controller.cpp:171-180; the reference
 * passes 4.0 m as the search distance, :172), for B
 * robots: closest pose of the not yet pruned plan within `max_robot_pose_search_dist` of integrated path length (:56-66),
 * poses from there up to the first one farther than `dist_threshold` from the robot (:68-75; the reference passes half
 * the larger costmap side), moved into the costmap frame (:77-96) and the plan pruned up to the closest pose (:98).
 * The reference's tf2 lookups are the caller's: `robot_pose` is the robot in the plan frame, `to_local` the rigid
 * transform plan frame -> costmap frame per robot (NULL: the frames coincide). nav2_util's euclidean_distance /
 * first_after_integrated_distance / min_by are restated in their ROS 2 Humble form (unpinned by the reference). */
enum smpc_window_error {
  SMPC_WINDOW_OK = 0,
  SMPC_WINDOW_EMPTY_PLAN = 1,   /* "Received plan with zero length" (:44-47): nothing left of the plan */
  SMPC_WINDOW_EMPTY_WINDOW = 2  /* "Resulting plan has 0 poses in it." (:100-103); the plan is pruned all the same */
};

typedef struct smpc_plan_window_batch {
  int32_t B;
  int32_t L;          /* row stride of `plan` and of the output window in poses */
  int32_t on_device;  /* 0: host pointers, 1: device pointers; outputs follow */
  int32_t reserved;
  double max_robot_pose_search_dist;
  double dist_threshold;
  const double* plan;        /* [B][L][2] global plans as stored by setPlan (:110-113), plan frame */
  const int32_t* plan_len;   /* [B] */
  int32_t* plan_start;       /* [B] in / out: poses already erased by earlier calls (global_plan_.poses.erase, :98) */
  const double* robot_pose;  /* [B][3] x, y, yaw in the plan frame */
  const double* to_local;    /* [B][3] tx, ty, yaw of the plan frame in the costmap frame, or NULL */
} smpc_plan_window_batch;

/* window [B][L][2] (the first window_len[b] poses of row b are written), window_len [B], error [B] (enum
 * smpc_window_error; may be NULL). The window is what smpc_trajectorize_path_batch takes as its plan. */
int smpc_transform_global_plan_batch(smpc_handle* h, const smpc_plan_window_batch* in, double* window, int32_t* window_len,
                                     int32_t* error);

/* The command SocialMPCController::computeVelocityCommands returns (src/social_mpc_controller.cpp:171-256; SURVEY §8 row
 * f4) for B robots: cmds[0] of a usable solve (:250-256), the trajectorizer's first command when the optimisation was
 * not usable (:241-245), (0.1, 0) when trajectorize returned false (:180-189), and nothing at all when
 * transformGlobalPlan threw (src/path_handler.cpp:44-47, 100-103: the exception leaves computeVelocityCommands and the
 * controller server publishes no command for the cycle). source [B]: 0 optimised, 1 trajectorizer command, 2 the
 * 0.1 m/s fallback, 3 no command (cmd_vel is written as zeros). A path shorter than T + 1 poses is not a fallback case:
 * it is solved with its own horizon (smpc_format_batch.n_poses -> T_scene). */
int smpc_select_command_batch(smpc_handle* h, int32_t B, int32_t T, int32_t traj_rows, int32_t on_device,
                              const int32_t* traj_n_poses /* [B]; NULL: every path is complete */,
                              const double* traj_cmds /* [B][traj_rows][2] */, const int32_t* status /* [B] */,
                              const double* cmds /* [B][T+1][2] */, double* cmd_vel /* [B][2] */,
                              int32_t* source /* [B]; may be NULL */,
                              const int32_t* window_error /* [B] enum smpc_window_error of this cycle; may be NULL */);

/* Diagnostic: evaluates the elementary functions the sweep uses (csrc/smpc_math.hpp: table-driven exp / atan2 /
 * sincos, refined reciprocal / rsqrt, and the raw hardware estimates behind them) on n host-side arguments, so that
 * tests can check them on the device against libm. fn: 0 exp(a) | 1 atan2(a, b) | 2 sin(a) -> out0, cos(a) -> out1 |
 * 3 1/sqrt(a) | 4 a / b | 5 raw v_rcp_f64(a) | 6 raw v_rsq_f64(a) | 7 the line search's bracketed root finder on n
 * quartics (a holds 8 doubles per problem: c4 c3 c2 c1 c0 lo hi pad; root -> out0, loop trips -> out1) | 8 atan2(a, b)
 * of a unit vector (the social force's angle: node table in LDS, no division).
 * out1 may be NULL unless fn == 2 or 7. */
int smpc_math_probe(smpc_handle* h, int32_t fn, int32_t n, const double* a, const double* b, double* out0, double* out1);

/* Diagnostic: measured FP64 vector peak of the device in TFLOP/s (independent v_fma_f64 chains, `iters` x 64 fused
 * multiply-adds per lane, 8 waves per SIMD, best of three launches) — the denominator of the FP64-VALU roofline
 * bench.py reports. Returns < 0 on error. */
double smpc_fp64_peak_probe(smpc_handle* h, int32_t iters);

/* Timing of the most recent kernel launched by this handle, measured with HIP events on the handle's
 * stream. Returns milliseconds, <0 if unavailable. Synchronises the stream. */
double smpc_last_kernel_ms(smpc_handle* h);

const char* smpc_last_error(void);
int smpc_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* SMPC_H_ */
