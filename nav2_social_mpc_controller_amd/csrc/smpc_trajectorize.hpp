// smpc_trajectorize.hpp — SURVEY §8 row f3 on the device: PathTrajectorizer::trajectorize
// (reference src/path_trajectorizer.cpp:120-288, motion model path_trajectorizer.hpp:106-135) for B plans.
// A pure-pursuit simulation: sequential in the step index, so one 16-lane group walks one plan, four plans per
// wavefront. Two kernels:
//  - smpc_trajectorize_kernel<kR> (plans of up to 16 kR <= 512 poses): the plan is read from HBM once and stays in
//    registers (kR poses per lane, pose i in lane (Lp-1-i) % 16), the look-ahead search of a step is kR branch-free
//    squared-distance tests per lane + one 16-lane minimum; heading / bearing use the table functions of
//    smpc_math.hpp; the steps' raw outputs are parked in LDS and written (with the quaternion round trip of the yaw)
//    by all lanes after the walk.
//    Longer plans take the same kernel after one pass that keeps the poses within reach of the start pose (only
//    those can ever lie inside the look-ahead circle) in scan order.
//  - smpc_trajectorize_long_kernel (more steps than the LDS park holds): the plan is searched where it lies, 64 poses
//    per trip from its end, library functions.
// Both keep the reference's scan order (first pose met inside the look-ahead circle when walking from the end of the
// plan, else the closest one with the first-met tie rule).
// Third-party arithmetic: angles::normalize_angle (ros/angles, version unpinned by the reference's package.xml) is
// restated in its ROS 2 form fmod(a + pi, 2 pi) -+ pi.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "smpc_math.hpp"

namespace smpc {

struct TrajParams {
  int B, L, max_steps, omnidirectional;
  int compact;  // register kernel: keep only the poses within reach of the start pose (plans longer than 16 kR poses)
  double desired_linear_vel, lookahead_dist, max_angular_vel, time_step;
  const double* plan;        // [B][L][2]
  const int32_t* plan_len;   // [B]
  const double* robot_pose;  // [B][3]
  double* path;              // [B][max_steps+1][3]
  double* cmds;              // [B][max_steps+1][2]
  double* cmds_vy;           // [B][max_steps+1] or null
  int32_t* n_poses;          // [B]
  int32_t* error;            // [B] or null
  MathTab mt;
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
__global__ __launch_bounds__(64) void smpc_trajectorize_long_kernel(const TrajParams p) {
#pragma clang fp contract(off)  // distances decide the way-point: keep them the plain IEEE products and sums of the reference
  SMPC_CHAIN_PRIORITY();
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

// ---- register-resident plans ----------------------------------------------------------------------------------------
constexpr int kTrajParkDoubles = 6;  // per step: x, y, raw heading, vx, wz, vy
constexpr int kTrajBlock = 256;      // four wavefronts per workgroup: one per SIMD of a CU, so 2 x 256 CUs x 4 SIMDs take 8192 plans evenly

// normalize_angle of an atan2 result a in [-pi, pi]: a + pi lies in [0, 2 pi], where fmod(., 2 pi) is the identity
// except at 2 pi itself
__device__ inline double traj_normalize_atan(double a) {
  double r = a + M_PI;
  r = (r >= 2.0 * M_PI) ? r - 2.0 * M_PI : r;
  return (r <= 0.0) ? r + M_PI : r - M_PI;
}

__device__ inline double traj_atan2(MathTabP mt, double y, double x) {
  // the table routine wants a non-degenerate vector; both-zero / tiny / huge operands take the library's cases
  const double m = fmax(fabs(x), fabs(y));
  if (!(m > 1e-100 && m < 1e100)) return atan2(y, x);
  return atan2_dir(mt, y, x);
}

// The sine / cosine table of the walk, copied out of the kernel-argument segment once: the loop has two wavefronts per
// SIMD to hide latency with, and two scalar-memory round trips per step for coefficients were a fifth of its time.
struct TrajSinCos {
  double sin_c[7], cos_c[8], two_over_pi, pio2_hi, pio2_lo;
};
__device__ inline void traj_pin(double& v) { asm volatile("" : "+s"(v)); }  // an opaque value in an SGPR pair
__device__ inline void traj_load_sincos(MathTabP mt, TrajSinCos* t) {
#pragma unroll
  for (int j = 0; j < 7; ++j) { t->sin_c[j] = mt->sin_c[j]; traj_pin(t->sin_c[j]); }
#pragma unroll
  for (int j = 0; j < 8; ++j) { t->cos_c[j] = mt->cos_c[j]; traj_pin(t->cos_c[j]); }
  t->two_over_pi = mt->two_over_pi; traj_pin(t->two_over_pi);
  t->pio2_hi = mt->pio2_hi; traj_pin(t->pio2_hi);
  t->pio2_lo = mt->pio2_lo; traj_pin(t->pio2_lo);
}

template <class TabP>
__device__ inline void traj_sincos(TabP mt, double a, double* sn, double* cs) {
  if (!(fabs(a) <= 1e5)) { *sn = sin(a); *cs = cos(a); return; }
  sincos_tab(mt, a, sn, cs);
}

// minimum over the 16 lanes of a group (= one DPP row) by four row rotations: VALU latency, no LDS crossbar trip
__device__ inline int traj_row_min(int v) {
  v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x128, 0xF, 0xF, false));  // row_ror:8
  v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x124, 0xF, 0xF, false));  // row_ror:4
  v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x122, 0xF, 0xF, false));  // row_ror:2
  v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x121, 0xF, 0xF, false));  // row_ror:1
  return v;
}

// The two rare searches of the register kernel read the plan where it lies (so the hot loop carries no code for them).
// Both return this lane's candidate as key = 16 u + gl (pose Lp-1-key), 0x7fffffff for none.
// (a) some pose lies within 1e-12 of the circle: the reference's comparison sqrt(z) <= look, first pose met.
__device__ __noinline__ int traj_first_inside_exact(const double* plan, int Lp, double rx, double ry, double look,
                                                    double look2_lo, double look2_hi, int gl) {
#pragma clang fp contract(off)
  for (int i = Lp - 1 - gl; i >= 0; i -= kTrajGroup) {
    const double wx = plan[2 * i], wy = plan[2 * i + 1];
    const double z = (rx - wx) * (rx - wx) + (ry - wy) * (ry - wy);
    if (z < look2_lo || (z <= look2_hi && sqrt(z) <= look)) return Lp - 1 - i;
  }
  return 0x7fffffff;
}
// (b) no pose inside the circle: the running minimum over sqrt distances from 100 m down, strict, so that the
// first-met pose keeps a tie (:167-174); reduced over the 16 lanes of the group.
__device__ __noinline__ int traj_closest(const double* plan, int Lp, double rx, double ry, int gl) {
#pragma clang fp contract(off)
  double dmin = 100.0;
  int kmin = 0x7fffffff;
  for (int i = Lp - 1 - gl; i >= 0; i -= kTrajGroup) {
    const double wx = plan[2 * i], wy = plan[2 * i + 1];
    double d = sqrt((rx - wx) * (rx - wx) + (ry - wy) * (ry - wy));
    d = (d == d) ? d : __builtin_inf();
    if (d < dmin) { dmin = d; kmin = Lp - 1 - i; }
  }
#pragma unroll
  for (int off = kTrajGroup / 2; off >= 1; off >>= 1) {
    const double od = __shfl_xor(dmin, off, 64);
    const int ok = __shfl_xor(kmin, off, 64);
    if (od < dmin || (od == dmin && ok < kmin)) { dmin = od; kmin = ok; }
  }
  return kmin;
}

template <int kR>
__global__ __launch_bounds__(kTrajBlock) void smpc_trajectorize_kernel(const TrajParams) {
#pragma clang fp contract(off)  // distances decide the way-point: keep them the plain IEEE products and sums of the reference
  SMPC_CHAIN_PRIORITY();
  extern __shared__ double traj_park[];  // [4 groups][max_steps][kTrajParkDoubles]
  const auto& p = *(const TrajParams __attribute__((address_space(4)))*)__builtin_amdgcn_kernarg_segment_ptr();
  const MathTabP mt = &p.mt;
  const int lane = threadIdx.x & 63;
  const int grp = lane / kTrajGroup, gl = lane - grp * kTrajGroup;  // group inside the wavefront
  const int bgrp = threadIdx.x / kTrajGroup;                        // group inside the block (its LDS park)
  const int scene = blockIdx.x * ((int)blockDim.x / kTrajGroup) + bgrp;  // blocks of kTrajBlock or of one wavefront
  if (scene >= p.B) return;
  const size_t s = scene;
  const int S1 = p.max_steps + 1;
  double* out_path = p.path + s * S1 * 3;
  double* out_cmds = p.cmds + s * S1 * 2;
  double* out_vy = p.cmds_vy ? p.cmds_vy + s * S1 : nullptr;
  double* park = traj_park + (size_t)bgrp * p.max_steps * kTrajParkDoubles;
  const int Lp = p.plan_len[s];
  int err = 0, steps = 0;
  const double rx0 = p.robot_pose[3 * s], ry0 = p.robot_pose[3 * s + 1], rth0 = p.robot_pose[3 * s + 2];
  if (Lp < 2 || Lp > p.L) {
    err = 1;  // "Path has less than 2 poses, cannot trajectorize" (:123-127): returns false
  } else {
    const double* plan = p.plan + s * (size_t)p.L * 2;
    // pose Lp-1 - (16 u + gl) in slot u: walking u upwards, and lanes upwards inside a slot, is the reference's scan
    // from the end of the plan. Slots past the start of the plan hold +inf (never inside the circle, never closest).
    double px[kR], py[kR];
    const double look = p.lookahead_dist, look2 = look * look;
    const double look2_lo = look2 * (1.0 - 1e-12), look2_hi = look2 * (1.0 + 1e-12);
    // Only a pose within look + max_steps |v| dt of the start pose can ever lie inside the look-ahead circle (the robot
    // travels at most |v| dt per step).
    const double reach = (fabs(look) + fabs(p.desired_linear_vel * p.time_step) * p.max_steps) * (1.0 + 1e-6) + 1e-9;
    const double reach2 = reach * reach;
    bool overflow = false;
    if (!p.compact) {
#pragma unroll
      for (int u = 0; u < kR; ++u) {
        const int i = Lp - 1 - (u * kTrajGroup + gl);
        const double2 w = (i >= 0) ? *(const double2*)(plan + 2 * i) : make_double2(__builtin_inf(), __builtin_inf());
        px[u] = w.x; py[u] = w.y;
      }
    } else {
      // A plan longer than the register file: one pass from its end keeps the reachable poses, in scan order, in an
      // LDS list (the order is all the search needs: "first met from the end" = smallest list position); the slots
      // are filled from the list. More reachable poses than slots: this plan's search reads the plan in memory.
      double2* list = reinterpret_cast<double2*>(traj_park + (size_t)(blockDim.x / kTrajGroup) * p.max_steps * kTrajParkDoubles) +
                      (size_t)bgrp * (kR * kTrajGroup);
      int count = 0;
      constexpr int kU = 8;  // 8 x 16 poses per trip: all loads of a trip are in flight before the first ballot
      for (int base = Lp - 1; base >= 0; base -= kU * kTrajGroup) {
        double2 w[kU];
#pragma unroll
        for (int v = 0; v < kU; ++v) {
          const int i = base - v * kTrajGroup - gl;
          w[v] = (i >= 0) ? *(const double2*)(plan + 2 * i) : make_double2(__builtin_nan(""), __builtin_nan(""));
        }
#pragma unroll
        for (int v = 0; v < kU; ++v) {
          const double z0 = (rx0 - w[v].x) * (rx0 - w[v].x) + (ry0 - w[v].y) * (ry0 - w[v].y);
          const bool in = base - v * kTrajGroup - gl >= 0;
          const bool cand = in && !(z0 > reach2);
          const unsigned m = (unsigned)((__ballot(cand) >> (grp * kTrajGroup)) & 0xFFFFull);
          const int pos = count + __popc(m & ((1u << gl) - 1u));
          if (cand && pos < kR * kTrajGroup) list[pos] = w[v];
          count += __popc(m);
        }
      }
      overflow = count > kR * kTrajGroup;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      const int kept = min(count, kR * kTrajGroup);
#pragma unroll
      for (int u = 0; u < kR; ++u) {
        const int j = u * kTrajGroup + gl;
        const double2 w = (j < kept) ? list[j] : make_double2(__builtin_inf(), __builtin_inf());
        px[u] = w.x; py[u] = w.y;
      }
    }
    double rx = rx0, ry = ry0, rth = rth0;
    const double gx = plan[2 * (Lp - 1)], gy = plan[2 * (Lp - 1) + 1];
    // Slots that can ever hold a pose inside the circle: one bit per slot, set when any lane of the wavefront holds a
    // reachable pose there: the search of a step only visits those slots (2-3 of 25 for a 20 m plan).
    uint32_t near_slots = 0;
    {
#pragma unroll
      for (int u = 0; u < kR; ++u) {
        const double z0 = (rx - px[u]) * (rx - px[u]) + (ry - py[u]) * (ry - py[u]);
        near_slots |= __ballot(!(z0 > reach2)) ? (1u << u) : 0u;  // (a NaN bound or pose keeps the slot in the search)
      }
      asm volatile("" : "+s"(near_slots));  // one mask in one SGPR (not kR booleans in SGPR pairs): s_bitcmp1 per slot
    }
    TrajSinCos sct;
    traj_load_sincos(mt, &sct);
    bool far_from_goal = true;
    while (far_from_goal && steps < p.max_steps) {
      // --- 1: look-ahead point (:160-175). usel = slot of the first pose met inside the circle in this lane, once
      // with the circle shrunk and once grown by 1e-12 (relative): where the two agree the lane's answer does not
      // depend on the rounding of the distance (so it is taken with a fused multiply-add); where they differ the
      // group repeats the search with the reference's expression, sqrt included.
      int usel = 64, usel_hi = 64;
      double hx = 0.0, hy = 0.0;  // this lane's first pose inside the circle
      uint32_t near = near_slots;
      asm volatile("" : "+s"(near));  // tested bit by bit here, not hoisted into kR loop-invariant flags
#pragma unroll
      for (int u = kR - 1; u >= 0; --u) {
        if ((near >> u) & 1u) {
          const double ax = rx - px[u], ay = ry - py[u];
          const double z = fma(ax, ax, ay * ay);
          const bool inside = z < look2_lo;
          usel = inside ? u : usel;
          hx = inside ? px[u] : hx;
          hy = inside ? py[u] : hy;
          usel_hi = (z <= look2_hi) ? u : usel_hi;
        }
      }
      int key = usel < 64 ? usel * kTrajGroup + gl : 0x7fffffff;
      const bool exact = overflow || (__ballot(usel != usel_hi) >> (grp * kTrajGroup) & 0xFFFFull) != 0;
      if (exact) key = traj_first_inside_exact(plan, Lp, rx, ry, look, look2_lo, look2_hi, gl);
      key = traj_row_min(key);
      double wpx, wpy;
      if (key != 0x7fffffff) {
        // the way-point is the first hit of one lane of the group: that lane holds its coordinates
        const int src = grp * kTrajGroup + (key & (kTrajGroup - 1));
        wpx = __shfl(hx, src, 64); wpy = __shfl(hy, src, 64);
        if (exact) { wpx = plan[2 * (Lp - 1 - key)]; wpy = plan[2 * (Lp - 1 - key) + 1]; }
      } else {
        key = traj_closest(plan, Lp, rx, ry, gl);
        if (key == 0x7fffffff) { err = 2; break; }  // every pose farther than 100 m: the reference reads poses[-1]
        wpx = plan[2 * (Lp - 1 - key)]; wpy = plan[2 * (Lp - 1 - key) + 1];
      }
      // --- 2: way-point in the robot frame, control law (:180-225)
      double sn, cs;
      traj_sincos(&sct, rth, &sn, &cs);
      const double dx = (wpx - rx) * cs + (wpy - ry) * sn;
      const double dy = -(wpx - rx) * sn + (wpy - ry) * cs;
      // dtheta = normalize_angle(atan2(dy, dx)) enters the control law only through its direction (omnidirectional) or
      // through |dtheta| > pi/2 and its sign (differential): both are read off (dx, dy) without the arctangent unless
      // the vector lies within 1e-15 rad of an axis that decides (then the reference's expression is evaluated).
      double vx = 0.0, vy = 0.0, wz = 0.0;
      const double point_dist2 = dx * dx + dy * dy;
      if (p.omnidirectional) {
        double sd, cd;
        if (point_dist2 > 1e-200 && point_dist2 < 1e200) {
          const double inv = rsqrt_pos(point_dist2);
          cd = dx * inv; sd = dy * inv;
        } else {
          traj_sincos(mt, traj_normalize_atan(traj_atan2(mt, dy, dx)), &sd, &cd);
        }
        vx = p.desired_linear_vel * cd;
        vy = p.desired_linear_vel * sd;
      } else {
        double curvature = 0.0;
        if (point_dist2 > 0.001) curvature = div_fast(2.0 * dy, point_dist2);
        vx = p.desired_linear_vel;
        bool behind, left;  // |dtheta| > pi/2, dtheta > 0
        if (dx > 0.0) {
          behind = false; left = false;
        } else if (dx < 0.0 && -dx > 1e-15 * fabs(dy) && fabs(dy) > 1e-15 * -dx) {
          behind = true; left = dy > 0.0;
        } else {
          const double dtheta = traj_normalize_atan(traj_atan2(mt, dy, dx));
          behind = fabs(dtheta) > M_PI / 2.0; left = dtheta > 0;
        }
        if (behind) {  // rotate in place
          vx = 0.0;
          wz = p.max_angular_vel * (left ? 1.0 : -1.0);
        } else {
          wz = vx * curvature;
        }
      }
      // --- 3: motion model (path_trajectorizer.hpp:106-135)
      double tx = vx * cs, ty = vx * sn;
      if (p.omnidirectional) {
        double sq, cq;
        traj_sincos(&sct, M_PI_2 + rth, &sq, &cq);
        tx = tx + vy * cq; ty = ty + vy * sq;
      }
      rx = rx + tx * p.time_step;
      ry = ry + ty * p.time_step;
      rth = rth + wz * p.time_step;
      if (gl == 0) {
        double* o = park + (size_t)steps * kTrajParkDoubles;
        o[0] = rx; o[1] = ry; o[2] = rth; o[3] = vx; o[4] = wz; o[5] = vy;
      }
      // goal_dist > 0.2 (:147): decided on the squared distance, the sqrt taken only within 1e-12 of the threshold
      const double g2 = (rx - gx) * (rx - gx) + (ry - gy) * (ry - gy);
      far_from_goal = g2 > 0.04 * (1.0 + 1e-12) || (g2 >= 0.04 * (1.0 - 1e-12) && sqrt(g2) > 0.2);
      ++steps;
    }
  }
  // --- write-out by the whole group: parked rows (yaw through setRPY -> toMsg -> getYaw, :232-240), zeros beyond
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  const int n_poses = err == 1 ? 0 : steps + 1;
  for (int k = gl; k < S1; k += kTrajGroup) {
    double ox = 0.0, oy = 0.0, oth = 0.0;
    if (k == 0 && n_poses > 0) { ox = rx0; oy = ry0; oth = rth0; }  // new_path.poses[0] = robot_pose (:137)
    if (k >= 1 && k < n_poses) {
      const double* o = park + (size_t)(k - 1) * kTrajParkDoubles;
      ox = o[0]; oy = o[1];
      double sz, cz;
      traj_sincos(mt, o[2] * 0.5, &sz, &cz);
      oth = traj_atan2(mt, 2.0 * (cz * sz), cz * cz - sz * sz);
    }
    out_path[3 * k] = ox; out_path[3 * k + 1] = oy; out_path[3 * k + 2] = oth;
    double cv = 0.0, cw = 0.0, cy = 0.0;
    if (k < steps) {
      const double* o = park + (size_t)k * kTrajParkDoubles;
      cv = o[3]; cw = o[4]; cy = o[5];
    }
    out_cmds[2 * k] = cv; out_cmds[2 * k + 1] = cw;
    if (out_vy) out_vy[k] = cy;
  }
  if (gl == 0) {
    p.n_poses[s] = n_poses;
    if (p.error) p.error[s] = err;
  }
}

}  // namespace smpc
