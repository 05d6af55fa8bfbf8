// smpc_project.hpp — batched people projection: Optimizer::project_people + computeObstacle
// (reference src/optimizer.cpp:554-728) with the Social Force Model of include/nav2_social_mpc_controller/sfm.hpp
// (computeForces :462-485 = desired + obstacle + social force, group forces identically zero here; updatePosition
// :525-551). One lane per agent (the robot is the last lane of the group), G = next power of two >= N+1 lanes per
// copy of a scene; while 2 G <= 64 a scene is held twice in the wavefront and the two copies split the partner rounds of
// the social force between them (everything else they compute identically, so no state has to be exchanged besides the
// two partial sums): the dependent chain of a step is what the kernel's time is made of, and this halves its longest
// part. Other agents' states travel by wavefront shuffles; the T steps are sequential.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/smpc.h"
#include "smpc_math.hpp"

namespace smpc {

struct ProjParams {
  int B, T, N, G;  // G = lanes of one copy of a scene's agents (power of two >= N + 1)
  int H;           // copies of a scene in the wavefront (2 when 2 G <= 64, else 1): each copy evaluates its share of the partner rounds
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

// computeObstacle (src/optimizer.cpp:673-728): nearest-obstacle lookup, float arithmetic as in the reference; the
// result is agent - obstacle (the reference stores this DIFFERENCE where the SFM expects a position). In two parts so
// that the grid load of a step is in flight during the next step's desired and social forces: `issue` finds the cell
// and loads its entry, `finish` (called where the obstacle force needs it) turns the entry into the difference.
template <typename PP>
__device__ inline int proj_obstacle_issue(const PP& p, const uint32_t* idx, double ox, double oy, double px, double py,
                                          unsigned int& ob) {
  const double res = (double)p.od_resolution;
  const unsigned int xcell = (unsigned int)(long long)floor((px - ox) / res);
  const unsigned int ycell = (unsigned int)(long long)floor((py - oy) / res);
  ob = 0;
  if (xcell >= (unsigned int)p.od_width || ycell >= (unsigned int)p.od_height) return SMPC_PROJ_CELL_OUT_OF_BOUNDS;
  ob = idx[xcell + ycell * (unsigned int)p.od_width];
  return SMPC_PROJ_OK;
}
template <typename PP>
__device__ inline int proj_obstacle_finish(const PP& p, unsigned int ob, double ox, double oy, double px, double py,
                                           double& dx, double& dy) {
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
  const int G = p.G, H = p.H, T = p.T, N = p.N;
  // a scene owns H x G lanes: H copies of its agents (copy hh = 0 writes the outputs), lane g of a copy = agent g
  const int Gs = G * H;
  const int grp = lane / Gs, hh = (lane - grp * Gs) / G, g = lane - grp * Gs - hh * G, base = grp * Gs + hh * G;
  const int scene_raw = blockIdx.x * (64 / Gs) + grp;
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
  if (live_scene && hh == 0)
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
  int err = SMPC_PROJ_OK, ob_err = SMPC_PROJ_OK;
  unsigned int ob = 0;  // grid entry of the agent's cell, loaded at the end of the previous step
  if (is_agent) {
    px = init[src * 6]; py = init[src * 6 + 1]; yaw = init[src * 6 + 2]; lv = init[src * 6 + 4]; av = init[src * 6 + 5];
    double sn, cs;
    sincos(yaw, &sn, &cs);
    vx = lv * cs; vy = lv * sn;
    des = 0.5; radius = 0.5;
    gx = px + (double)p.max_time * vx; gy = py + (double)p.max_time * vy; has_goal = true;  // constant-velocity goal :588-592
    ob_err = proj_obstacle_issue(p, idx, ox, oy, px, py, ob);
  }
  // the robot's row of the coming step is loaded one step ahead (its latency would sit on every step's critical path)
  double nr0 = 0, nr1 = 0, nr2 = 0, nr4 = 0, nr5 = 0, goal_x = 0, goal_y = 0;
  if (is_robot) {
    nr0 = rpath[0]; nr1 = rpath[1]; nr2 = rpath[2]; nr4 = rpath[4]; nr5 = rpath[5];
    goal_x = rpath[(size_t)T * 6]; goal_y = rpath[(size_t)T * 6 + 1];
  }
  for (int i = 0; i < T; ++i) {
    if (is_robot) {  // the robot re-enters from the initial trajectory every step (:613-630)
      px = nr0; py = nr1; yaw = nr2; lv = nr4; av = nr5;
      const double* r = rpath + (size_t)min(i + 1, T - 1) * 6;
      nr0 = r[0]; nr1 = r[1]; nr2 = r[2]; nr4 = r[4]; nr5 = r[5];
      double sn, cs;
      if (__builtin_expect(!(fabs(yaw) <= 1e5), 0)) sincos(yaw, &sn, &cs);
      else sincos_tab(mt, yaw, &sn, &cs);
      vx = lv * cs; vy = lv * sn;
      des = 0.6; radius = 0.5;
      gx = goal_x; gy = goal_y; has_goal = true;
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
    // computeSocialForce(index, agents) :237-281. The force on i from j is the exact negative of the force on j from i
    // (diff, velocity difference and with them the interaction vector flip sign; theta, B and both exponentials are
    // unchanged), so every unordered pair is evaluated once: in round k lane i takes partner (i + k) mod n and hands the
    // negated force to it by shuffle; for even n the last round (k = n/2) pairs lanes mutually and needs no hand-over.
    // The sum over partners runs in round order instead of index order (differences at round-off level).
    // Two rounds per trip: their evaluations are independent chains, so each lane keeps two in flight (the kernel runs
    // with two wavefronts per SIMD and is bound by the latency of one chain otherwise).
    // Copy hh of the scene takes the rounds 1 + hh, 1 + hh + H, ...; the copies' sums are added below.
    double ax = 0.0, ay = 0.0;  // this copy's share of the social force
    for (int kk = 1 + hh; 2 * kk <= n_act; kk += 2 * H) {
      const bool two = 2 * (kk + H) <= n_act;
      const bool act = g < n_act;
      double sfx[2], sfy[2];
      ProjPair q[2];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int k = (h == 1 && !two) ? kk : kk + h * H;
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
      if (act) { ax += sfx[0]; ay += sfy[0]; }
      if (2 * kk != n_act) {
        const int src = (g - kk + n_act) % n_act;
        const double rx = __shfl(sfx[0], base + src, 64), ry = __shfl(sfy[0], base + src, 64);
        if (act) { ax -= rx; ay -= ry; }
      }
      if (two) {
        if (act) { ax += sfx[1]; ay += sfy[1]; }
        if (2 * (kk + H) != n_act) {
          const int src = (g - kk - H + 2 * n_act) % n_act;
          const double rx = __shfl(sfx[1], base + src, 64), ry = __shfl(sfy[1], base + src, 64);
          if (act) { ax -= rx; ay -= ry; }
        }
      }
    }
    if (H == 2) {  // both copies form the same sum (copy 0's share + copy 1's share), so they stay bit-identical
      const double ox_ = __shfl_xor(ax, G, 64), oy_ = __shfl_xor(ay, G, 64);
      ax = (hh == 0) ? ax + ox_ : ox_ + ax;
      ay = (hh == 0) ? ay + oy_ : oy_ + ay;
    }
    fx += ax; fy += ay;
    if (is_agent) {  // computeObstacleForce :207-235, one obstacle entry, used as a POSITION
      // (after the social force in program order: the grid entry loaded at the end of the last step arrives meanwhile)
      int oe = ob_err;
      if (!oe) oe = proj_obstacle_finish(p, ob, ox, oy, px, py, obx, oby);
      if (oe && !err) err = oe;
      const double mx = px - obx, my = py - oby;
      const double z = mx * mx + my * my;
      const double inv = rsqrt_pos(fmax(z, 1e-300));
      const double mn = z * inv;
      const double e = kFo * exp_tab(mt, -(mn - radius) * (1.0 / kSig));
      fx += e * (z > 0 ? mx * inv : mx);
      fy += e * (z > 0 ? my * inv : my);
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
    if (is_agent) ob_err = proj_obstacle_issue(p, idx, ox, oy, px, py, ob);
    if (live_scene && hh == 0 && g < N) {
      double* o = out + (size_t)(i + 1) * 6 * N + g;
      if (is_agent) {
        o[0] = px; o[N] = py; o[2 * N] = yaw; o[3 * N] = (double)((float)(i + 1) * p.time_step); o[4 * N] = lv; o[5 * N] = av;
      } else {
        o[0] = 0.0; o[N] = 0.0; o[2 * N] = 0.0; o[3 * N] = -1.0; o[4 * N] = 0.0; o[5 * N] = 0.0;
      }
    }
  }
  if (is_agent) {  // the entry refreshed after the last step is checked like the others (:636-640)
    int e = ob_err;
    if (!e) e = proj_obstacle_finish(p, ob, ox, oy, px, py, obx, oby);
    if (e && !err) err = e;
  }
  // per-scene error: any agent lane that hit a grid exception
  int any = err;
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    const int other = __shfl_xor(any, off, 64);
    if (off < Gs && other && !any) any = other;
  }
  if (live_scene && g == 0 && hh == 0 && p.error) p.error[s] = any;
}

}  // namespace smpc
