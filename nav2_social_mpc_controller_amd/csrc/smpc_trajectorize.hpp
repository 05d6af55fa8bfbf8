// smpc_trajectorize.hpp — SURVEY §8 row f3 on the device: PathTrajectorizer::trajectorize
// (reference src/path_trajectorizer.cpp:120-288, motion model path_trajectorizer.hpp:106-135) for B plans.
// A pure-pursuit simulation: sequential in the step index, so one 16-lane group walks one plan; the look-ahead search
// over the plan poses (the only O(L) part of a step) is spread over the group's lanes, 16 poses per trip from the end
// of the plan, with the reference's scan order kept by ballots. Four plans per wavefront.
// Third-party arithmetic: angles::normalize_angle (ros/angles, version unpinned by the reference's package.xml) is
// restated in its ROS 2 form fmod(a + pi, 2 pi) -+ pi.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace smpc {

struct TrajParams {
  int B, L, max_steps, omnidirectional;
  double desired_linear_vel, lookahead_dist, max_angular_vel, time_step;
  const double* plan;        // [B][L][2]
  const int32_t* plan_len;   // [B]
  const double* robot_pose;  // [B][3]
  double* path;              // [B][max_steps+1][3]
  double* cmds;              // [B][max_steps+1][2]
  double* cmds_vy;           // [B][max_steps+1] or null
  int32_t* n_poses;          // [B]
  int32_t* error;            // [B] or null
};

constexpr int kTrajGroup = 16;

__device__ inline double traj_normalize_angle(double a) {
  const double r = fmod(a + M_PI, 2.0 * M_PI);
  return (r <= 0.0) ? r + M_PI : r - M_PI;
}

__device__ inline double traj_yaw_roundtrip(double yaw) {  // setRPY(0, 0, yaw) -> toMsg -> tf2::getYaw
  double sz, cz;
  sincos(yaw * 0.5, &sz, &cz);
  return atan2(2.0 * (cz * sz), cz * cz - sz * sz);
}

// The plan is searched where it lies (L2-resident after the first step): staging it in LDS was measured slower (0.36 vs
// 0.25 ms for 8192 plans of 400 poses) because 4 x L x 16 bytes of LDS per wavefront cost more occupancy than the
// latency they save.
__global__ __launch_bounds__(64) void smpc_trajectorize_kernel(const TrajParams p) {
#pragma clang fp contract(off)  // distances decide the way-point: keep them the plain IEEE products and sums of the reference
  const int lane = threadIdx.x & 63;
  const int grp = lane / kTrajGroup, gl = lane - grp * kTrajGroup;
  const int scene = blockIdx.x * (64 / kTrajGroup) + grp;
  if (scene >= p.B) return;
  const size_t s = scene;
  const int S1 = p.max_steps + 1;
  double* out_path = p.path + s * S1 * 3;
  double* out_cmds = p.cmds + s * S1 * 2;
  double* out_vy = p.cmds_vy ? p.cmds_vy + s * S1 : nullptr;
  const int Lp = p.plan_len[s];
  int err = 0, steps = 0;
  if (Lp < 2 || Lp > p.L) {
    err = 1;  // "Path has less than 2 poses, cannot trajectorize" (:123-127): returns false
  } else {
    const double* plan = p.plan + s * (size_t)p.L * 2;
    double rx = p.robot_pose[3 * s], ry = p.robot_pose[3 * s + 1], rth = p.robot_pose[3 * s + 2];
    if (gl == 0) { out_path[0] = rx; out_path[1] = ry; out_path[2] = rth; }  // new_path.poses[0] = robot_pose (:137)
    const double gx = plan[2 * (Lp - 1)], gy = plan[2 * (Lp - 1) + 1];
    const unsigned shift = grp * kTrajGroup;
    const double look2 = p.lookahead_dist * p.lookahead_dist;
    const double look2_lo = look2 * (1.0 - 1e-12), look2_hi = look2 * (1.0 + 1e-12);
    double goal_dist = 1000.0;
    while (goal_dist > 0.2 && steps < p.max_steps) {
      // --- 1: look-ahead point, scanning from the end of the plan (:160-175): the first pose met inside the look-ahead
      // circle, else the closest one. Pass A only looks for a hit (squared distances; the sqrt of the reference is
      // taken only for a pose within 1e-12 of the circle, where it could decide); pass B — no pose of the plan inside
      // the circle, rare — is the reference's running minimum over sqrt distances with its first-met tie rule.
      int wp_index = -1;
      constexpr int kU = 4;  // 4 x 16 poses per trip: all loads of a trip are in flight before the first compare
      for (int base = Lp - 1; base >= 0 && wp_index < 0; base -= kU * kTrajGroup) {
        double z[kU];
        bool in[kU];
#pragma unroll
        for (int u = 0; u < kU; ++u) {
          const int i = base - u * kTrajGroup - gl;
          in[u] = i >= 0;
          const int ic = in[u] ? i : 0;
          const double wx = plan[2 * ic], wy = plan[2 * ic + 1];
          z[u] = (rx - wx) * (rx - wx) + (ry - wy) * (ry - wy);
        }
#pragma unroll
        for (int u = 0; u < kU; ++u) {
          bool hit = in[u] && z[u] < look2_lo;
          const bool maybe = in[u] && !hit && z[u] <= look2_hi;
          if ((__ballot(maybe) >> shift) & 0xFFFFull) hit = hit || (maybe && sqrt(z[u]) <= p.lookahead_dist);
          const unsigned hits = (unsigned)((__ballot(hit) >> shift) & 0xFFFFu);
          if (hits && wp_index < 0) wp_index = base - u * kTrajGroup - (__ffs(hits) - 1);
        }
      }
      if (wp_index < 0) {
        double min_dist = 100.0;
        for (int base = Lp - 1; base >= 0; base -= kTrajGroup) {
          const int i = base - gl;
          const bool in = i >= 0;
          double d = __builtin_inf();
          if (in) {
            const double wx = plan[2 * i], wy = plan[2 * i + 1];
            d = sqrt((rx - wx) * (rx - wx) + (ry - wy) * (ry - wy));
            d = (d == d) ? d : __builtin_inf();
          }
          double m = d;
#pragma unroll
          for (int off = kTrajGroup / 2; off >= 1; off >>= 1) m = fmin(m, __shfl_xor(m, off, 64));
          if (m < min_dist) {  // strict: an earlier trip (later plan poses) keeps a tie
            const unsigned eq = (unsigned)((__ballot(in && d == m) >> shift) & 0xFFFFu);
            min_dist = m;
            wp_index = base - (__ffs(eq) - 1);
          }
        }
      }
      if (wp_index < 0) { err = 2; break; }  // every pose farther than 100 m: the reference reads poses[-1]
      const double wpx = plan[2 * wp_index], wpy = plan[2 * wp_index + 1];
      // --- 2: way-point in the robot frame, control law (:180-225)
      double sn, cs;
      sn = sin(rth); cs = cos(rth);
      const double dx = (wpx - rx) * cs + (wpy - ry) * sn;
      const double dy = -(wpx - rx) * sn + (wpy - ry) * cs;
      const double dtheta = traj_normalize_angle(atan2(dy, dx));
      double vx = 0.0, vy = 0.0, wz = 0.0;
      if (p.omnidirectional) {
        vx = p.desired_linear_vel * cos(dtheta);
        vy = p.desired_linear_vel * sin(dtheta);
      } else {
        const double point_dist2 = dx * dx + dy * dy;
        double curvature = 0.0;
        if (point_dist2 > 0.001) curvature = 2.0 * dy / point_dist2;
        vx = p.desired_linear_vel;
        if (fabs(dtheta) > M_PI / 2.0) {  // rotate in place
          vx = 0.0;
          wz = p.max_angular_vel * (dtheta > 0 ? 1.0 : -1.0);
        } else {
          wz = vx * curvature;
        }
      }
      // --- 3: motion model (path_trajectorizer.hpp:106-135)
      double tx = vx * cs, ty = vx * sn;
      if (p.omnidirectional) { tx = tx + vy * cos(M_PI_2 + rth); ty = ty + vy * sin(M_PI_2 + rth); }
      rx = rx + tx * p.time_step;
      ry = ry + ty * p.time_step;
      rth = rth + wz * p.time_step;
      if (gl == 0) {
        double* o = out_path + 3 * (steps + 1);
        o[0] = rx; o[1] = ry; o[2] = traj_yaw_roundtrip(rth);
        out_cmds[2 * steps] = vx; out_cmds[2 * steps + 1] = wz;
        if (out_vy) out_vy[steps] = vy;
      }
      goal_dist = sqrt((rx - gx) * (rx - gx) + (ry - gy) * (ry - gy));
      ++steps;
    }
  }
  // rows the simulation did not reach are zero
  const int n_poses = err == 1 ? 0 : steps + 1;
  for (int k = gl; k < S1; k += kTrajGroup) {
    if (k >= n_poses) { out_path[3 * k] = 0.0; out_path[3 * k + 1] = 0.0; out_path[3 * k + 2] = 0.0; }
    if (k >= steps) {
      out_cmds[2 * k] = 0.0; out_cmds[2 * k + 1] = 0.0;
      if (out_vy) out_vy[k] = 0.0;
    }
  }
  if (gl == 0) {
    p.n_poses[s] = n_poses;
    if (p.error) p.error[s] = err;
  }
}

}  // namespace smpc
