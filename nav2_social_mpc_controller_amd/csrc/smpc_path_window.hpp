// smpc_path_window.hpp — SURVEY §8 row f4, the part of the plugin shell that is geometry rather than ROS plumbing:
// mpc::PathHandler::transformGlobalPlan (reference src/path_handler.cpp:39-108) for B robots — the window of the global
// plan that computeVelocityCommands hands to the trajectorizer (src/social_mpc_controller.cpp:171-180) and the pruning
// of the poses the robot has passed. One wavefront per plan:
//   A  first_after_integrated_distance (:56-59): segment lengths by all lanes, the running sum in the reference's own
//      order (a uniform loop over the lanes' values: the comparison with the bound sees the very sums of the reference);
//   B  min_by over [start, upper) (:61-66): lane-strided strict minima, then the smallest index among equal minima;
//   C  first pose from there farther than the threshold (:68-75): 64 poses per ballot;
//   D  the window, through the rigid transform plan frame -> costmap frame (:77-96; the reference's tf2 lookups are the
//      caller's: it passes (tx, ty, yaw) per robot, or nothing for identity); the plan's start index moves up (:98).
// Third-party arithmetic restated from nav2_util/geometry_utils.hpp (ROS 2 Humble; unpinned by the reference):
// euclidean_distance = hypot(dx, dy), first_after_integrated_distance, min_by.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/smpc.h"
#include "smpc_math.hpp"

namespace smpc {

struct WindowParams {
  int B, L;
  double search_dist, dist_threshold;
  const double* plan;        // [B][L][2]
  const int32_t* plan_len;   // [B]
  int32_t* plan_start;       // [B] in / out
  const double* robot_pose;  // [B][3] in the plan frame
  const double* to_local;    // [B][3] (tx, ty, yaw) or null
  double* window;            // [B][L][2]
  int32_t* window_len;       // [B]
  int32_t* error;            // [B] or null
};

__global__ __launch_bounds__(64) void smpc_plan_window_kernel(const WindowParams p) {
#pragma clang fp contract(off)  // distances decide indices: keep the plain products and sums of the reference
  SMPC_CHAIN_PRIORITY();
  const int lane = threadIdx.x & 63;
  const int sidx = blockIdx.x;
  if (sidx >= p.B) return;
  const size_t s = sidx;
  const double* plan = p.plan + s * (size_t)p.L * 2;
  double* out = p.window + s * (size_t)p.L * 2;
  const int n = min(max(p.plan_len[s], 0), p.L);
  const int start = max(p.plan_start[s], 0);
  if (n - start <= 0) {  // "Received plan with zero length" (:44-47)
    if (lane == 0) { p.window_len[s] = 0; if (p.error) p.error[s] = SMPC_WINDOW_EMPTY_PLAN; }
    return;
  }
  const double rx = p.robot_pose[3 * s], ry = p.robot_pose[3 * s + 1];
  // ---- A: upper bound of the closest-pose search
  int upper = n;
  {
    double dist = 0.0;
    bool found = false;
    for (int base = start; base < n - 1 && !found; base += 64) {
      const int i = base + lane;
      double d = 0.0;
      if (i < n - 1) d = hypot(plan[2 * (i + 1)] - plan[2 * i], plan[2 * (i + 1) + 1] - plan[2 * i + 1]);
      const int cnt = min(64, n - 1 - base);
      for (int k = 0; k < cnt; ++k) {  // the reference's running sum, element by element
        dist += __shfl(d, k, 64);
        if (dist > p.search_dist) { upper = base + k + 1; found = true; break; }
      }
    }
  }
  // ---- B: closest pose in [start, upper): first minimum
  double dmin = __builtin_inf();
  int imin = 0x7fffffff;
  for (int i = start + lane; i < upper; i += 64) {
    const double d = hypot(rx - plan[2 * i], ry - plan[2 * i + 1]);
    // (a NaN distance a lane met first must not shadow the finite ones it meets later: min_by only lets a NaN stick
    // when it is the first element of the WHOLE range, which the rule behind the reduction below restores)
    if (imin == 0x7fffffff || d < dmin || (dmin != dmin && d == d)) { dmin = d; imin = i; }
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    const double od = __shfl_xor(dmin, off, 64);
    const int oi = __shfl_xor(imin, off, 64);
    // a NaN distance never wins over a number; among equal distances the earlier pose stays (min_by's strict <)
    const bool take = (oi != 0x7fffffff) && (imin == 0x7fffffff || od < dmin || (od == dmin && oi < imin) || (dmin != dmin && od == od));
    if (take) { dmin = od; imin = oi; }
  }
  // min_by starts from the first element and only moves on a strict <: a NaN distance there stays the "lowest"
  const double d_first = hypot(rx - plan[2 * start], ry - plan[2 * start + 1]);
  const int tb = (imin == 0x7fffffff || d_first != d_first || dmin != dmin) ? start : imin;
  // ---- C: first pose from tb on that is farther than the threshold
  int te = n;
  for (int base = tb; base < n; base += 64) {
    const int i = base + lane;
    const bool far = i < n && hypot(plan[2 * i] - rx, plan[2 * i + 1] - ry) > p.dist_threshold;
    const unsigned long long m = __ballot(far);
    if (m) { te = base + (__ffsll((long long)m) - 1); break; }
  }
  // ---- D: the window in the costmap frame, the pruned start
  double tx = 0.0, ty = 0.0, c = 1.0, sn = 0.0;
  if (p.to_local) {
    tx = p.to_local[3 * s]; ty = p.to_local[3 * s + 1];
    const double yaw = p.to_local[3 * s + 2];
    c = cos(yaw); sn = sin(yaw);
  }
  for (int i = tb + lane; i < te; i += 64) {
    const double x = plan[2 * i], y = plan[2 * i + 1];
    double ox = x, oy = y;
    if (p.to_local) { ox = tx + c * x - sn * y; oy = ty + sn * x + c * y; }
    out[2 * (i - tb)] = ox; out[2 * (i - tb) + 1] = oy;
  }
  if (lane == 0) {
    p.plan_start[s] = tb;                       // global_plan_.poses.erase(begin, transformation_begin) (:98)
    p.window_len[s] = max(te - tb, 0);
    if (p.error) p.error[s] = (te - tb <= 0) ? SMPC_WINDOW_EMPTY_WINDOW : SMPC_WINDOW_OK;  // (:100-103)
  }
}

}  // namespace smpc
