// smpc_format.hpp — SURVEY §8 row f2 on the device: Optimizer::format_to_optimize (reference src/optimizer.cpp:484-551)
// and the TrajectoryMemory store (:448-449) for B scenes, plus the solve inputs Optimizer::optimize derives from the
// formatted status (:197-261). Elementwise work: one lane per (scene, pose), no cross-lane traffic; HBM-bound
// (reads 2 x (3 + 2) doubles, writes 6 + 2 (+ 5 into an empty memory record) doubles per pose).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "smpc_math.hpp"

namespace smpc {

struct FormatParams {
  int B, T, nb, P, rows;  // rows: pose stride of path / cmds
  int max_poses;          // round(max_time / time_step) of the cut (:492-497); 0: no cut
  float time_step, current_path_w, current_cmds_w;
  const double* path;   // [B][rows][3]
  const double* cmds;   // [B][rows][2]
  const double* speed;  // [B][2]
  const int32_t* n_poses;  // [B] poses of each incoming path, or null: every path has (at least) T + 1
  double* prev_path;    // [B][T+1][3]
  double* prev_cmds;    // [B][T+1][2]
  int32_t* valid;       // [B]
  int32_t* length;      // [B][2] poses / commands a record holds, or null (fixed horizon: T + 1 of both)
  double* robot_status; // [B][T+1][6]
  double* pose0;        // [B][3]
  double* init_params;  // [B][P]
  double* path_pts;     // [B][T+1][2]
  double* goal_yaw;     // [B]
  int32_t* T_scene;     // [B] or null
};

// tf2 Quaternion::setRPY(0, 0, yaw) -> toMsg -> tf2::getYaw for a pure-yaw quaternion (x = y = 0).
__device__ inline double format_yaw_roundtrip(double yaw) {
  double sz, cz;
  sincos(yaw * 0.5, &sz, &cz);
  return atan2(2.0 * (cz * sz), cz * cz - sz * sz);
}

// poses of scene s as they arrive (n), poses format_to_optimize keeps (kept: the cut of :492-497, at most the stride)
__device__ inline void format_lengths(const FormatParams& p, int s, int& n, int& kept) {
  const int Tp = p.T + 1;
  n = p.n_poses ? max(p.n_poses[s], 0) : Tp;
  kept = (p.max_poses > 0 && n > p.max_poses) ? p.max_poses - 1 : n;
  kept = min(kept, Tp);
}

// grid: ceil(B * (T + 1) / 256) blocks of 256 lanes; lane = (scene, pose index i)
__global__ __launch_bounds__(256) void smpc_format_kernel(const FormatParams p) {
  SMPC_CHAIN_PRIORITY();
  const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int Tp = p.T + 1;
  if (gid >= (long long)p.B * Tp) return;
  const int s = (int)(gid / Tp), i = (int)(gid - (long long)s * Tp);
  const size_t e = (size_t)s * Tp + i;          // element of the dense [B][T+1] arrays
  const size_t ei = (size_t)s * p.rows + i;      // element of the incoming path / cmds
  int n, kept;
  format_lengths(p, s, n, kept);
  const int ncmd = p.n_poses ? max(n - 1, 0) : Tp;  // commands that arrive: one per step taken (trajectorize :262-269)
  const bool have = p.valid[s] != 0;
  // memory.previous_path.poses.size() == 0: previous := current, the whole incoming path and commands (:177-183; as
  // much of them as the record's T + 1 rows hold); the blend below then runs against that copy
  const int plen = have ? (p.length ? p.length[2 * s] : Tp) : min(n, Tp);
  const int clen = have ? (p.length ? p.length[2 * s + 1] : Tp) : min(ncmd, Tp);
  if (!have) {
    if (i < plen) { p.prev_path[3 * e] = p.path[3 * ei]; p.prev_path[3 * e + 1] = p.path[3 * ei + 1]; p.prev_path[3 * e + 2] = p.path[3 * ei + 2]; }
    if (i < clen) { p.prev_cmds[2 * e] = p.cmds[2 * ei]; p.prev_cmds[2 * e + 1] = p.cmds[2 * ei + 1]; }
  }
  double* r = p.robot_status + 6 * e;
  if (i == 0 && p.T_scene) p.T_scene[s] = max(kept - 1, 0);  // optim_velocities.size() after the pop_back (:237)
  if (i >= kept) {  // rows the scene does not have: defined content
    r[0] = r[1] = r[2] = r[3] = r[4] = r[5] = 0.0;
    p.path_pts[2 * e] = 0.0; p.path_pts[2 * e + 1] = 0.0;
    if (i < p.nb) { p.init_params[(size_t)s * p.P + 2 * i] = 0.0; p.init_params[(size_t)s * p.P + 2 * i + 1] = 0.0; }
    if (kept == 0 && i == 0) { p.pose0[3 * s] = p.pose0[3 * s + 1] = p.pose0[3 * s + 2] = 0.0; p.goal_yaw[s] = 0.0; }
    return;
  }
  const double cx = p.path[3 * ei], cy = p.path[3 * ei + 1], cyaw = p.path[3 * ei + 2];
  const double wp = (double)p.current_path_w, wc = (double)p.current_cmds_w;
  double x = cx, y = cy, yaw = cyaw;
  if (i < plen) {  // "!previous_path.poses.empty() && i < previous_path.poses.size()" (:504): blended, yaw through setRPY
    const double px = have ? p.prev_path[3 * e] : cx, py = have ? p.prev_path[3 * e + 1] : cy;
    const double pyaw = have ? p.prev_path[3 * e + 2] : cyaw;
    x = wp * cx + (1.0 - wp) * px;
    y = wp * cy + (1.0 - wp) * py;
    yaw = format_yaw_roundtrip(wp * cyaw + (1.0 - wp) * pyaw);
  }
  double lv, av;
  if (i == 0) {
    lv = p.speed[2 * s]; av = p.speed[2 * s + 1];  // :529-533
  } else {
    // cmds[i - 1] against previous_cmds[i - 1] (:537-545). The reference indexes previous_cmds without a bound: a
    // record shorter than i (a plan that grew by more than a pose since the last solve) is undefined behaviour there;
    // here the current command stands alone in that case.
    const size_t em = e - 1;
    const double cv1 = p.cmds[2 * (ei - 1)], cw1 = p.cmds[2 * (ei - 1) + 1];
    const bool in = i - 1 < clen;
    const double pv1 = (have && in) ? p.prev_cmds[2 * em] : cv1, pw1 = (have && in) ? p.prev_cmds[2 * em + 1] : cw1;
    lv = wc * cv1 + (1.0 - wc) * pv1;
    av = wc * cw1 + (1.0 - wc) * pw1;
  }
  r[0] = x; r[1] = y; r[2] = yaw;
  r[3] = (double)((float)i * p.time_step);  // unsigned * float product (:523)
  r[4] = lv; r[5] = av;
  // what Optimizer::optimize takes from optim_status (:206-235, 254-261, 298)
  p.path_pts[2 * e] = x; p.path_pts[2 * e + 1] = y;
  if (i == 0) {
    p.pose0[3 * s] = x; p.pose0[3 * s + 1] = y;
    p.pose0[3 * s + 2] = format_yaw_roundtrip(yaw);  // evolving_poses[0]: setRPY(0, 0, yaw), read back with getYaw
  }
  if (i < p.nb) { p.init_params[(size_t)s * p.P + 2 * i] = lv; p.init_params[(size_t)s * p.P + 2 * i + 1] = av; }
  if (i == kept - 1) p.goal_yaw[s] = yaw;
}

// Second pass of the format step: mark freshly filled memory records valid (after every lane of the first kernel has
// read the flag) and note how much they hold.
__global__ __launch_bounds__(256) void smpc_format_mark_kernel(const FormatParams p) {
  SMPC_CHAIN_PRIORITY();
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= p.B || p.valid[s] != 0) return;
  int n, kept;
  format_lengths(p, s, n, kept);
  p.valid[s] = 1;
  if (p.length) {
    p.length[2 * s] = min(n, p.T + 1);
    p.length[2 * s + 1] = min(p.n_poses ? max(n - 1, 0) : p.T + 1, p.T + 1);
  }
}

struct PeopleParams {
  int B, Np, N;
  const double* people;   // [B][Np][5]
  const int32_t* count;   // [B]
  double* init_people;    // [B][N][6]
  uint8_t* has_people;    // [B] or null
  // field-of-view filter (robot_pose == null: off)
  const double* robot_pose;      // [B][3]
  double fov_angle;
  const double* costmap_origin;  // [B or 1][2]
  int costmap_shared, size_x, size_y;
  double resolution;
};

// angles::shortest_angular_distance(from, to) = normalize_angle(to - from) (ros/angles, ROS 2 form)
__device__ inline double people_shortest_angular_distance(double from, double to) {
  const double r = fmod((to - from) + M_PI, 2.0 * M_PI);
  return (r <= 0.0) ? r + M_PI : r - M_PI;
}

// The field-of-view filter of computeVelocityCommands (src/social_mpc_controller.cpp:196-214) followed by
// Optimizer::people_to_status (src/optimizer.cpp:454-482): one lane per scene, the persons of a scene are walked in
// order (the filter compacts, people_to_status truncates to the first N and pads with t = -1).
__global__ __launch_bounds__(256) void smpc_people_to_status_kernel(const PeopleParams p) {
  SMPC_CHAIN_PRIORITY();
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= p.B) return;
  const int cnt = min(max(p.count[s], 0), p.Np);
  double* o = p.init_people + (size_t)s * p.N * 6;
  const bool filter = p.robot_pose != nullptr;
  double rx = 0, ry = 0, ox = 0, oy = 0;
  float robot_yaw = 0.f;
  if (filter) {
    rx = p.robot_pose[3 * s]; ry = p.robot_pose[3 * s + 1];
    robot_yaw = (float)p.robot_pose[3 * s + 2];
    ox = p.costmap_origin[p.costmap_shared ? 0 : 2 * s]; oy = p.costmap_origin[p.costmap_shared ? 1 : 2 * s + 1];
  }
  int kept = 0;
  for (int a = 0; a < cnt; ++a) {
    const double* q = p.people + ((size_t)s * p.Np + a) * 5;
    const double px = q[0], py = q[1];
    if (filter) {
      // Costmap2D::worldToMap (nav2_costmap_2d): inside the map <=> not left of / below the origin and cell < size
      if (px < ox || py < oy) continue;
      const unsigned mx = (unsigned)(long long)((px - ox) / p.resolution), my = (unsigned)(long long)((py - oy) / p.resolution);
      if (!(mx < (unsigned)p.size_x && my < (unsigned)p.size_y)) continue;
      const float angle_to_person = (float)atan2(py - ry, px - rx);
      const float relative = (float)people_shortest_angular_distance((double)robot_yaw, (double)angle_to_person);
      if (!((double)fabsf(relative) < p.fov_angle)) continue;
    }
    if (kept < p.N) {
      const double vx = q[2], vy = q[3];
      double* r = o + (size_t)kept * 6;
      r[0] = px; r[1] = py; r[2] = atan2(vy, vx); r[3] = 0.0; r[4] = sqrt(vx * vx + vy * vy); r[5] = q[4];
    }
    ++kept;
  }
  for (int k = min(kept, p.N); k < p.N; ++k) {  // "we fill with invalid agent: time=-1" (:470-476)
    double* r = o + (size_t)k * 6;
    r[0] = 0.0; r[1] = 0.0; r[2] = 0.0; r[3] = -1.0; r[4] = 0.0; r[5] = 0.0;
  }
  if (p.has_people) p.has_people[s] = kept != 0 ? 1 : 0;
}

struct StoreParams {
  int B, T;
  const int32_t* status;
  const double* path;  // [B][T+1][3] smpc_result_batch.path
  const double* cmds;  // [B][T+1][2] smpc_result_batch.cmds
  const int32_t* T_scene;  // [B] or null: every scene has T steps
  double* prev_path;
  double* prev_cmds;
  int32_t* valid;
  int32_t* length;     // [B][2] or null
};

__global__ __launch_bounds__(256) void smpc_memory_store_kernel(const StoreParams p) {
  SMPC_CHAIN_PRIORITY();
  const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int Tp = p.T + 1;
  if (gid >= (long long)p.B * Tp) return;
  const int s = (int)(gid / Tp), i = (int)(gid - (long long)s * Tp);
  if (p.status[s] == 2 /* SMPC_FAILURE */) return;  // the reference returns false before the store (:384-388)
  const int Tb = p.T_scene ? min(max(p.T_scene[s], 1), p.T) : p.T;  // the path and the commands of a solve have Tb + 1 entries
  if (i > Tb) return;
  const size_t e = (size_t)gid;
  p.prev_path[3 * e] = p.path[3 * e]; p.prev_path[3 * e + 1] = p.path[3 * e + 1]; p.prev_path[3 * e + 2] = p.path[3 * e + 2];
  p.prev_cmds[2 * e] = p.cmds[2 * e]; p.prev_cmds[2 * e + 1] = p.cmds[2 * e + 1];
  if (i == 0) {
    p.valid[s] = 1;
    if (p.length) { p.length[2 * s] = Tb + 1; p.length[2 * s + 1] = Tb + 1; }
  }
}

struct SelectParams {
  int B, T, rows;
  const int32_t* traj_n;   // [B] or null
  const double* traj_cmds; // [B][rows][2]
  const int32_t* status;   // [B]
  const double* cmds;      // [B][T+1][2]
  const int32_t* window_error;  // [B] or null
  double* cmd_vel;         // [B][2]
  int32_t* source;         // [B] or null
};

// computeVelocityCommands' choice of the returned command (src/social_mpc_controller.cpp:171-189, 241-245, 250-256).
__global__ __launch_bounds__(256) void smpc_select_command_kernel(const SelectParams p) {
  SMPC_CHAIN_PRIORITY();
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= p.B) return;
  const int n = p.traj_n ? p.traj_n[s] : p.T + 1;
  int src;
  double v, w;
  if (p.window_error && p.window_error[s] != 0) {
    // transformGlobalPlan threw (src/path_handler.cpp:44-47, 100-103): computeVelocityCommands does not return, the
    // controller server gets the exception and no command goes out in this cycle
    src = 3; v = 0.0; w = 0.0;
  } else if (n <= 0) {  // trajectorize() returned false: "using fallback cmd_vel" (:180-189)
    src = 2; v = 0.1; w = 0.0;
  } else if (p.status[s] == 2 /* SMPC_FAILURE */) {  // optimize() returned false: cmds = init_cmds (:241-245)
    src = 1; v = p.traj_cmds[(size_t)s * p.rows * 2]; w = p.traj_cmds[(size_t)s * p.rows * 2 + 1];
  } else {
    src = 0; v = p.cmds[(size_t)s * (p.T + 1) * 2]; w = p.cmds[(size_t)s * (p.T + 1) * 2 + 1];
  }
  p.cmd_vel[2 * s] = v; p.cmd_vel[2 * s + 1] = w;
  if (p.source) p.source[s] = src;
}

}  // namespace smpc
