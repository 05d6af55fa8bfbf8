// smpc_project.hpp — batched people projection: Optimizer::project_people + computeObstacle
// (reference src/optimizer.cpp:554-728) with the Social Force Model of include/nav2_social_mpc_controller/sfm.hpp
// (computeForces :462-485 = desired + obstacle + social force, group forces identically zero here; updatePosition
// :525-551). One lane per agent (the robot is the last lane of the group), G = next power of two >= N+1 lanes per
// scene, 64/G scenes per wavefront; other agents' states travel by wavefront shuffles; the T steps are sequential.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/smpc.h"
#include "smpc_math.hpp"

namespace smpc {

struct ProjParams {
  int B, T, N, G;
  float max_time, time_step, od_resolution;
  int od_shared, od_width, od_height;
  const double* init_people;
  const double* robot_path;
  const uint32_t* od_indexes;
  const double* od_origin;
  double* people_proj;
  int32_t* error;
  MathTab mt;  // polynomial tables of smpc_math.hpp (scalar loads, see there)
};

// The reference normalises angles by repeated +-2 pi (sfm angle.hpp); every argument on this path is a difference of
// atan2 results except `yaw - init_yaw` at step 0, where init_yaw is caller data. A wave must terminate whatever the
// input: beyond 8 pi the value is pre-reduced with fmod (non-finite input comes back as NaN), within it the loop is the
// reference's own arithmetic.
__device__ inline double proj_wrap(double a) {
  if (!(fabs(a) <= 8.0 * M_PI)) a = fmod(a, 2 * M_PI);
  while (a <= -M_PI) a += 2 * M_PI;
  while (a > M_PI) a -= 2 * M_PI;
  return a;
}

// computeObstacle (src/optimizer.cpp:673-728): nearest-obstacle lookup, float arithmetic as in the reference;
// returns agent - obstacle (the reference stores this DIFFERENCE where the SFM expects a position).
template <typename PP>
__device__ inline int proj_obstacle(const PP& p, const uint32_t* idx, double ox, double oy, double px, double py,
                                    double& dx, double& dy) {
  const double res = (double)p.od_resolution;
  const unsigned int xcell = (unsigned int)(long long)floor((px - ox) / res);
  const unsigned int ycell = (unsigned int)(long long)floor((py - oy) / res);
  if (xcell >= (unsigned int)p.od_width || ycell >= (unsigned int)p.od_height) return SMPC_PROJ_CELL_OUT_OF_BOUNDS;
  const unsigned int ob = idx[xcell + ycell * (unsigned int)p.od_width];
  if (ob >= (unsigned int)p.od_width * (unsigned int)p.od_height) return SMPC_PROJ_INDEX_OUT_OF_BOUNDS;
  const unsigned int oyc = ob / (unsigned int)p.od_width, oxc = ob % (unsigned int)p.od_width;
  const float x = (float)((double)((float)oxc * p.od_resolution) + ox);
  const float y = (float)((double)((float)oyc * p.od_resolution) + oy);
  dx = px - (double)x;
  dy = py - (double)y;
  return SMPC_PROJ_OK;
}

// sqrt(z) for z >= 0 through the refined reciprocal square root (1-2 ulp; 0 stays 0)
__device__ inline double proj_sqrt(double z) { return z > 0.0 ? z * rsqrt_pos(z) : 0.0; }

// computeSocialForce's term of one partner (sfm.hpp:237-281): diff = partner - self, dv = own velocity - partner's.
// Constants of the reference's defaults (forceFactorSocial 2.1, lambda 2, gamma 0.35, n 2, n' 3). Two phases, so that
// a caller with several partners in flight keeps their arithmetic in straight-line code and visits the rare
// two-arctangent form once for all of them.
struct ProjPair {
  double ex, ey, ix, iy, il, earg, theta;
  bool same_vel, near_axis;
};
__device__ inline void proj_pair_begin(MathTabP mt, double dfx, double dfy, double dvx, double dvy, ProjPair& q) {
  const double kLam = 2.0, kGam = 0.35;
  const double z = dfx * dfx + dfy * dfy;
  const double inv_nd = rsqrt_pos(fmax(z, 1e-300));
  const double nd = z * inv_nd;
  q.ex = z > 0 ? dfx * inv_nd : dfx; q.ey = z > 0 ? dfy * inv_nd : dfy;
  const double ivx = kLam * dvx + q.ex, ivy = kLam * dvy + q.ey;
  const double inv_il = rsqrt_pos(ivx * ivx + ivy * ivy);
  q.il = (ivx * ivx + ivy * ivy) * inv_il;
  q.ix = ivx * inv_il; q.iy = ivy * inv_il;
  // equal velocities (two standing people): theta is mathematically 0 and the reference gets its libm's last-bit
  // noise (its thetaSign is then 0 or +-1 by chance); take exactly 0, the convention of the hot path
  q.same_vel = (kLam * dvx == 0.0) && (kLam * dvy == 0.0);
  // theta = wrap(atan2(e) - atan2(i)) is the angle from i to e = atan2(i x e, i . e): one table arctangent away from
  // theta = 0 and |theta| = pi, the reference's own two-atan2 form next to them (its last bits decide thetaSign)
  const double cross = q.ix * q.ey - q.iy * q.ex, dot = q.ix * q.ex + q.iy * q.ey;
  q.theta = atan2_dir(mt, cross, dot);
  q.near_axis = !q.same_vel && !(fabs(cross) >= 1e-6);  // (equal velocities: i = e, cross = 0, theta := 0 anyway)
  q.earg = -nd * inv_il * (1.0 / kGam);  // -|diff| / B
}
__device__ inline void proj_pair_exact_theta(ProjPair& q) {
  if (q.near_axis) q.theta = proj_wrap(proj_wrap(atan2(q.ey, q.ex)) - proj_wrap(atan2(q.iy, q.ix)));
}
__device__ inline void proj_pair_end(MathTabP mt, const ProjPair& q, double& sfx, double& sfy) {
  const double kFs = 2.1, kGam = 0.35, kN = 2.0, kNp = 3.0;
  const double theta = q.same_vel ? 0.0 : q.theta;
  const double Bq = kGam * q.il;
  const double a1 = kNp * Bq * theta, a2 = kN * Bq * theta;
  const double fv = -exp_tab(mt, fma(-a1, a1, q.earg));
  const double sgn = (theta == 0) ? 0.0 : ((theta > 0) ? 1.0 : -1.0);  // sfm.hpp:265-270
  const double fa = -sgn * exp_tab(mt, fma(-a2, a2, q.earg));
  sfx = kFs * (fv * q.ix + fa * (-q.iy));
  sfy = kFs * (fv * q.iy + fa * q.ix);
}

__global__ __launch_bounds__(64) void smpc_project_kernel(const ProjParams) {
  SMPC_CHAIN_PRIORITY();
  const auto& p = *(const ProjParams __attribute__((address_space(4)))*)__builtin_amdgcn_kernarg_segment_ptr();
  const MathTabP mt = &p.mt;
  const int lane = threadIdx.x & 63;
  const int G = p.G, T = p.T, N = p.N;
  const int grp = lane / G, g = lane - grp * G, base = grp * G;
  const int scene_raw = blockIdx.x * (64 / G) + grp;
  const bool live_scene = scene_raw < p.B;
  const size_t s = live_scene ? scene_raw : p.B - 1;
  const double* init = p.init_people + s * (size_t)N * 6;
  const double* rpath = p.robot_path + s * (size_t)(T + 1) * 6;
  const uint32_t* idx = p.od_indexes + (p.od_shared ? 0 : s * (size_t)p.od_width * p.od_height);
  const double ox = p.od_origin[p.od_shared ? 0 : 2 * s], oy = p.od_origin[p.od_shared ? 1 : 2 * s + 1];
  double* out = p.people_proj + s * (size_t)(T + 1) * 6 * N;
  const double dt = (double)p.time_step;
  const bool grid_not_valid = (p.od_width == 100 && p.od_height == 100);  // src/optimizer.cpp:598-603
  const double kFd = 2.0, kFo = 20.0, kSig = 0.2, kRelax = 0.5;

  // people_traj[0] = init_people
  if (live_scene)
    for (int q = g; q < N * 6; q += G) { const int a = q / 6, f = q - a * 6; out[(size_t)f * N + a] = init[a * 6 + f]; }

  // compact the valid agents: lane g holds the g-th valid one; lane n_valid holds the robot
  int n_valid = 0, src = -1;
  for (int i = 0; i < N; ++i) {
    if (init[i * 6 + 3] == -1.0 || grid_not_valid) continue;
    if (n_valid == g) src = i;
    ++n_valid;
  }
  const bool is_agent = g < n_valid;
  const bool is_robot = g == n_valid;
  const int n_act = n_valid + 1;
  double px = 0, py = 0, vx = 0, vy = 0, yaw = 0, lv = 0, av = 0, des = 0.6, radius = 0.5;
  double gx = 0, gy = 0, grad = 0.25, obx = 0, oby = 0;
  bool has_goal = false;
  int err = SMPC_PROJ_OK;
  if (is_agent) {
    px = init[src * 6]; py = init[src * 6 + 1]; yaw = init[src * 6 + 2]; lv = init[src * 6 + 4]; av = init[src * 6 + 5];
    double sn, cs;
    sincos(yaw, &sn, &cs);
    vx = lv * cs; vy = lv * sn;
    des = 0.5; radius = 0.5;
    gx = px + (double)p.max_time * vx; gy = py + (double)p.max_time * vy; has_goal = true;  // constant-velocity goal :588-592
    const int e = proj_obstacle(p, idx, ox, oy, px, py, obx, oby);
    if (e) err = e;
  }
  for (int i = 0; i < T; ++i) {
    if (is_robot) {  // the robot re-enters from the initial trajectory every step (:613-630)
      const double* r = rpath + (size_t)i * 6;
      px = r[0]; py = r[1]; yaw = r[2]; lv = r[4]; av = r[5];
      double sn, cs;
      if (__builtin_expect(!(fabs(yaw) <= 1e5), 0)) sincos(yaw, &sn, &cs);
      else sincos_tab(mt, yaw, &sn, &cs);
      vx = lv * cs; vy = lv * sn;
      des = 0.6; radius = 0.5;
      gx = rpath[(size_t)T * 6]; gy = rpath[(size_t)T * 6 + 1]; has_goal = true;
    }
    // ---- computeForces (sfm.hpp:462-485)
    double fx, fy;
    {
      const double ddx = gx - px, ddy = gy - py;
      const double z = ddx * ddx + ddy * ddy;
      const double inv = rsqrt_pos(fmax(z, 1e-300));
      const double dn = z * inv;
      if (has_goal && dn > grad) {  // computeDesiredForce :188-205
        const double ux = z > 0 ? ddx * inv : ddx, uy = z > 0 ? ddy * inv : ddy;
        fx = kFd * (ux * des - vx) / kRelax;
        fy = kFd * (uy * des - vy) / kRelax;
      } else {
        fx = -vx / kRelax;
        fy = -vy / kRelax;
      }
    }
    if (is_agent) {  // computeObstacleForce :207-235, one obstacle entry, used as a POSITION
      const double mx = px - obx, my = py - oby;
      const double z = mx * mx + my * my;
      const double inv = rsqrt_pos(fmax(z, 1e-300));
      const double mn = z * inv;
      const double e = kFo * exp_tab(mt, -(mn - radius) * (1.0 / kSig));
      fx += e * (z > 0 ? mx * inv : mx);
      fy += e * (z > 0 ? my * inv : my);
    }
    // computeSocialForce(index, agents) :237-281. The force on i from j is the exact negative of the force on j from i
    // (diff, velocity difference and with them the interaction vector flip sign; theta, B and both exponentials are
    // unchanged), so every unordered pair is evaluated once: in round k lane i takes partner (i + k) mod n and hands the
    // negated force to it by shuffle; for even n the last round (k = n/2) pairs lanes mutually and needs no hand-over.
    // The sum over partners runs in round order instead of index order (differences at round-off level).
    // Two rounds per trip: their evaluations are independent chains, so each lane keeps two in flight (the kernel runs
    // with two wavefronts per SIMD and is bound by the latency of one chain otherwise).
    for (int kk = 1; 2 * kk <= n_act; kk += 2) {
      const bool two = 2 * (kk + 1) <= n_act;
      const bool act = g < n_act;
      double sfx[2], sfy[2];
      ProjPair q[2];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int k = (h == 1 && !two) ? kk : kk + h;
        const int j = (g + k) % n_act;
        const double qx = __shfl(px, base + j, 64), qy = __shfl(py, base + j, 64);
        const double wx = __shfl(vx, base + j, 64), wy = __shfl(vy, base + j, 64);
        proj_pair_begin(mt, qx - px, qy - py, vx - wx, vy - wy, q[h]);
      }
      q[0].near_axis = q[0].near_axis && act;  // idle lanes of the group carry no agent
      q[1].near_axis = q[1].near_axis && act;
      if (__builtin_expect(q[0].near_axis || q[1].near_axis, 0)) { proj_pair_exact_theta(q[0]); proj_pair_exact_theta(q[1]); }
#pragma unroll
      for (int h = 0; h < 2; ++h) proj_pair_end(mt, q[h], sfx[h], sfy[h]);
      if (act) { fx += sfx[0]; fy += sfy[0]; }
      if (2 * kk != n_act) {
        const int src = (g - kk + n_act) % n_act;
        const double rx = __shfl(sfx[0], base + src, 64), ry = __shfl(sfy[0], base + src, 64);
        if (act) { fx -= rx; fy -= ry; }
      }
      if (two) {
        if (act) { fx += sfx[1]; fy += sfy[1]; }
        if (2 * (kk + 1) != n_act) {
          const int src = (g - kk - 1 + n_act) % n_act;
          const double rx = __shfl(sfx[1], base + src, 64), ry = __shfl(sfy[1], base + src, 64);
          if (act) { fx -= rx; fy -= ry; }
        }
      }
    }
    // ---- updatePosition (sfm.hpp:525-551)
    vx += fx * dt; vy += fy * dt;
    {
      const double z = vx * vx + vy * vy;
      const double inv = rsqrt_pos(fmax(z, 1e-300));
      const double sp = z * inv;
      if (sp > des) { vx = (z > 0 ? vx * inv : vx) * des; vy = (z > 0 ? vy * inv : vy) * des; }
    }
    const double init_yaw = yaw;
    {  // atan2(vy, vx) lies in [-pi, pi] already: the wrap only moves -pi (vy = -0, vx < 0) to... itself + 2 pi = pi
      const double m = fmax(fabs(vx), fabs(vy));
      const double a = (m > 1e-100 && m < 1e100) ? atan2_dir(mt, vy, vx) : atan2(vy, vx);
      yaw = proj_wrap(a);
    }
    av = proj_wrap(yaw - init_yaw) / dt;
    px += vx * dt; py += vy * dt;
    lv = proj_sqrt(vx * vx + vy * vy);
    if (has_goal) {
      const double ddx = gx - px, ddy = gy - py;
      if (proj_sqrt(ddx * ddx + ddy * ddy) <= grad) has_goal = false;
    }
    // ---- refresh each person's obstacle entry (:636-640) and emit people_traj[i+1] (:642-668)
    if (is_agent) {
      const int e = proj_obstacle(p, idx, ox, oy, px, py, obx, oby);
      if (e && !err) err = e;
    }
    if (live_scene && g < N) {
      double* o = out + (size_t)(i + 1) * 6 * N + g;
      if (is_agent) {
        o[0] = px; o[N] = py; o[2 * N] = yaw; o[3 * N] = (double)((float)(i + 1) * p.time_step); o[4 * N] = lv; o[5 * N] = av;
      } else {
        o[0] = 0.0; o[N] = 0.0; o[2 * N] = 0.0; o[3 * N] = -1.0; o[4 * N] = 0.0; o[5 * N] = 0.0;
      }
    }
  }
  // per-scene error: any agent lane that hit a grid exception
  int any = err;
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    const int other = __shfl_xor(any, off, 64);
    if (off < G && other && !any) any = other;
  }
  if (live_scene && g == 0 && p.error) p.error[s] = any;
}

}  // namespace smpc
