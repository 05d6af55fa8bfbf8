// smpc_device.hpp — gfx950 device code of the batched social-MPC solver (one 64-lane wavefront per scene).
//
// What this restates, MI355X-first (reference files relative to /root/reference):
//   * rollout a1 (include/nav2_social_mpc_controller/update_state.hpp:37-63) as ONE shared rollout per sweep
//     plus closed-form sensitivities S_t = d(x,y,theta)_{t+1}/d(params) instead of the reference's per-functor
//     O(t) re-integration on ceres::Jet;
//   * the nine instantiated residual kinds a2..a9 (include/.../critics/*_cost_function.hpp) with analytic
//     state-space gradients (x, y, theta, v_block) chained with S_t — equal to Ceres autodiff in exact arithmetic;
//   * Gram contraction [J r]^T [J r] (gives J^T J, J^T r and the cost);
//   * the Ceres trust-region LM loop a11 and the post-solve unpack a12
//     (src/optimizer.cpp:390-446).
//
// Lane mapping of the sweep ("pair rounds"): the T*N (step, agent) social-force pairs are dealt to lanes in
// rounds of floor(64/N) whole steps; lane t (< T) then owns horizon step t for every per-step critic; the
// horizon pose block lives in LDS; per-step sums over agents go through LDS in a fixed order (deterministic).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/smpc.h"

namespace smpc {

constexpr int kWave = 64;
constexpr int kPairOut = 15;  // F(2) dF/d{x,y,th,v}(8) |F|^2(1) d|F|^2(4)

struct KParams {
  int B, T, N, CH, bl, nb, P, nbounded, nfeas;
  int size_x, size_y, costmap_shared;
  double dt, resolution;
  smpc_params prm;
  const double* pose0;
  const double* init_params;
  const double* path_pts;
  const double* goal_yaw;
  const double* people;
  const uint8_t* has_people;
  const uint8_t* costmap;
  const double* costmap_origin;
  // solve outputs
  double* o_params;
  double* o_cmds;
  double* o_path;
  int32_t* o_status;
  int32_t* o_reason;
  int32_t* o_iterations;
  int32_t* o_evaluations;
  double* o_initial_cost;
  double* o_final_cost;
  // eval (K1) inputs / outputs
  const double* e_x;
  double* e_residuals;
  double* e_jacobian;
  double* e_cost;
  double* e_gradient;
  int e_M;  // row stride of the eval outputs (M with people)
};

// LDS carve-up (in doubles) for one wave. Everything a sweep shares across lanes lives here.
struct LdsLayout {
  int ag;       // [5][T*N]   px, py, wx, wy, valid
  int pose;     // [5][T+1]   x, y, theta, cos, sin  (index k = pose after k steps)
  int pair;     // [15][64]
  int lm;       // LM vectors / matrices
  int scratch;  // polynomial scratch
  int total;
};

__host__ __device__ inline LdsLayout make_layout(int T, int N, int P) {
  LdsLayout L;
  int o = 0;
  L.ag = o; o += 5 * T * (N > 0 ? N : 1);
  L.pose = o; o += 5 * (T + 1);
  L.pair = o; o += kPairOut * kWave;
  L.lm = o; o += 12 * P + 2 * P * P + 8;
  L.scratch = o; o += 96;
  L.total = o;
  return L;
}

__device__ inline int lane_id() { return threadIdx.x & 63; }

__device__ inline double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}
__device__ inline int wave_any(int pred) { return __any(pred); }

__device__ inline double wrap_to_pi(double a) {  // critics/social_work_cost_function.hpp:39-46
  while (a > M_PI) a -= 2.0 * M_PI;
  while (a <= -M_PI) a += 2.0 * M_PI;
  return a;
}

// ------------------------------------------------------------------------------------------------
// Social force between one (me, other) pair and its Jacobians with respect to diff = me_pos - other_pos and
// u = me_vel - other_vel. Restates computeSocialForce (critics/social_work_cost_function.hpp:164-228) for a
// single "other"; constants from src/critics/social_work_cost_function.cpp:38-43.
// F = k (fv i + fa i_perp). Outputs F and the four directional derivatives along
//   d/d(diff_x), d/d(diff_y), d/d(u_x), d/d(u_y).
// ------------------------------------------------------------------------------------------------
struct Force {
  double fx, fy;
  double dfx_dx, dfy_dx, dfx_dy, dfy_dy;    // wrt diff
  double dfx_dux, dfy_dux, dfx_duy, dfy_duy;  // wrt u
};

__device__ inline Force social_force(double dx, double dy, double ux, double uy) {
  const double lambda = 2.0, gamma = 0.35, nPrime = 3.0, nn = 2.0, k = 2.1;
  Force R;
  double n2 = dx * dx + dy * dy;
  double n = sqrt(n2);
  bool degenerate = n < 1e-6;  // :181-184  diff := (1e-6, 0), a constant: no dependence on positions
  if (degenerate) { dx = 1e-6; dy = 0.0; n = sqrt(dx * dx + dy * dy); }
  const double inv_n = 1.0 / n;
  const double ex = dx * inv_n, ey = dy * inv_n;  // diffDirection
  const double ivx = lambda * ux + ex, ivy = lambda * uy + ey;  // :191-192
  const double L = sqrt(ivx * ivx + ivy * ivy);                 // :194
  const double inv_L = 1.0 / L;
  const double ix = ivx * inv_L, iy = ivy * inv_L;              // :195-196
  const double phi = wrap_to_pi(atan2(ey, ex) - atan2(iy, ix));  // :198-200
  const double Bq = gamma * L;                                  // :203
  const double inv_B = 1.0 / Bq;
  const double a1 = nPrime * Bq * phi, a2 = nn * Bq * phi;
  const double base = -n * inv_B;
  const double E1 = exp(base - a1 * a1);                        // :205-207
  const double E2 = exp(base - a2 * a2);                        // :212-215
  const double sgn = (phi > 0.0) ? 1.0 : -1.0;                  // :210
  const double fv = -E1, fa = -sgn * E2;
  // F = k (fv i + fa i_perp), i_perp = (-iy, ix)                  :218-224
  R.fx = k * (fv * ix - fa * iy);
  R.fy = k * (fv * iy + fa * ix);

  // directional derivative along a direction that moves: n by dn, alpha=atan2(e) by dalpha, iv by (divx, divy)
  auto dirderiv = [&](double dn, double dalpha, double divx, double divy, double& ofx, double& ofy) {
    const double dL = ix * divx + iy * divy;
    const double kappa = (-iy * divx + ix * divy) * inv_L;  // d beta = d atan2(i)
    const double dphi = dalpha - kappa;
    const double dB = gamma * dL;
    const double dbase = -dn * inv_B + n * dB * inv_B * inv_B;
    const double darg1 = dbase - 2.0 * a1 * nPrime * (dB * phi + Bq * dphi);
    const double darg2 = dbase - 2.0 * a2 * nn * (dB * phi + Bq * dphi);
    const double dfv = -E1 * darg1;
    const double dfa = -sgn * E2 * darg2;
    // dF = k ((dfv - fa kappa) i + (dfa + fv kappa) i_perp)
    const double ci = dfv - fa * kappa, cp = dfa + fv * kappa;
    ofx = k * (ci * ix - cp * iy);
    ofy = k * (ci * iy + cp * ix);
  };
  if (degenerate) {
    R.dfx_dx = R.dfy_dx = R.dfx_dy = R.dfy_dy = 0.0;
  } else {
    // dd = (1,0): dn = ex, dalpha = -ey/n, d(iv) = de = e_perp * dalpha, e_perp = (-ey, ex)
    double da = -ey * inv_n;
    dirderiv(ex, da, -ey * da, ex * da, R.dfx_dx, R.dfy_dx);
    da = ex * inv_n;  // dd = (0,1)
    dirderiv(ey, da, -ey * da, ex * da, R.dfx_dy, R.dfy_dy);
  }
  dirderiv(0.0, 0.0, lambda, 0.0, R.dfx_dux, R.dfy_dux);
  dirderiv(0.0, 0.0, 0.0, lambda, R.dfx_duy, R.dfy_duy);
  return R;
}

// ------------------------------------------------------------------------------------------------
// Bicubic interpolation of the u8 costmap with clamp-to-edge, value and gradient
// (ceres::BiCubicInterpolator<Grid2D<u_char>> semantics; used by
// critics/obstacle_cost_function.hpp:161 as Evaluate(row = y_cell, col = x_cell)).
// ------------------------------------------------------------------------------------------------
__device__ inline void cubic_hermite(double p0, double p1, double p2, double p3, double x, double& f, double& df) {
  const double a = 0.5 * (-p0 + 3.0 * p1 - 3.0 * p2 + p3);
  const double b = 0.5 * (2.0 * p0 - 5.0 * p1 + 4.0 * p2 - p3);
  const double c = 0.5 * (-p0 + p2);
  f = p1 + x * (c + x * (b + x * a));
  df = c + x * (2.0 * b + 3.0 * a * x);
}

__device__ inline void bicubic(const uint8_t* __restrict__ map, int size_x, int size_y, double r, double c,
                               double& f, double& dfdr, double& dfdc) {
  const double fr = floor(r), fc = floor(c);
  // keep the int conversion defined for wild coordinates; clamping below makes any far-outside index equivalent
  const double frc = fmin(fmax(fr, -4.0), (double)size_y + 4.0), fcc = fmin(fmax(fc, -4.0), (double)size_x + 4.0);
  const int row = (int)frc, col = (int)fcc;
  double fv[4], dv[4];
  int cc[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) cc[j] = min(max(col - 1 + j, 0), size_x - 1);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int rr = min(max(row - 1 + i, 0), size_y - 1);
    const uint8_t* p = map + (size_t)rr * size_x;
    cubic_hermite((double)p[cc[0]], (double)p[cc[1]], (double)p[cc[2]], (double)p[cc[3]], c - fc, fv[i], dv[i]);
  }
  double unused;
  cubic_hermite(fv[0], fv[1], fv[2], fv[3], r - fr, f, dfdr);
  cubic_hermite(dv[0], dv[1], dv[2], dv[3], r - fr, dfdc, unused);
}

// ------------------------------------------------------------------------------------------------
// Per-wave scene context
// ------------------------------------------------------------------------------------------------
template <int NB>
struct Ctx {
  static constexpr int P = 2 * NB;
  const KParams* kp;
  int scene;
  int lane;
  bool has_people;
  double x0, y0, yaw0, goal_yaw, ox, oy;
  const uint8_t* map;
  const double* path_pts;  // [T+1][2]
  double* lds;
  LdsLayout L;
  // per-lane (lane t owns step t): agent-angle tag (a7) and path targets
  bool aa_active;
  double aa_target;
  double tx, ty;    // path_pts[t+1]
  double gx, gy;    // final trajectorized point
  // pair slot of this lane within a round
  int spr;      // steps per round
  int pj, pa;   // step-in-round, agent index
  bool pslot;   // lane holds a pair slot
};

// Gram matrix [J r]^T [J r], packed upper triangle over P+1 columns (column P is r).
template <int P> struct Gram {
  static constexpr int Q = P + 1;
  static constexpr int SZ = Q * (Q + 1) / 2;
  double v[SZ];
  __device__ static constexpr int idx(int a, int b) { return a * Q - a * (a - 1) / 2 + (b - a); }  // a <= b
  __device__ inline void clear() {
#pragma unroll
    for (int i = 0; i < SZ; ++i) v[i] = 0.0;
  }
  __device__ inline void add_row(const double (&row)[P], double r) {
#pragma unroll
    for (int a = 0; a < P; ++a) {
#pragma unroll
      for (int b = a; b < P; ++b) v[idx(a, b)] = fma(row[a], row[b], v[idx(a, b)]);
      v[idx(a, P)] = fma(row[a], r, v[idx(a, P)]);
    }
    v[idx(P, P)] = fma(r, r, v[idx(P, P)]);
  }
  __device__ inline double H(int a, int b) const { return a <= b ? v[idx(a, b)] : v[idx(b, a)]; }
};

// One-time per-scene setup: stage people into LDS (px, py, wx, wy, valid), agent-angle tags, path targets.
template <int NB>
__device__ inline void setup_scene(Ctx<NB>& c) {
  const KParams& k = *c.kp;
  const int T = k.T, N = k.N, lane = c.lane;
  double* ag = c.lds + c.L.ag;
  const int TN = T * N;
  if (c.has_people) {
    const double* ppl = k.people + (size_t)c.scene * (T + 1) * 6 * N;
    for (int q = lane; q < TN; q += kWave) {
      const int t = q / N, a = q - t * N;
      const double* f = ppl + (size_t)(t + 1) * 6 * N + a;  // people_proj[t + 1]
      const double px = f[0], py = f[N], yaw = f[2 * N], tt = f[3 * N], lv = f[4 * N];
      double sn, cs;
      sincos(yaw, &sn, &cs);
      ag[0 * TN + q] = px;
      ag[1 * TN + q] = py;
      ag[2 * TN + q] = lv * cs;   // aVel, social_work:187-188
      ag[3 * TN + q] = lv * sn;
      ag[4 * TN + q] = (tt == -1.0) ? 0.0 : 1.0;  // :175
    }
  }
  // a7 AgentAngle tag: depends on constants only (critics/agent_angle_cost_function.hpp:130-190)
  c.aa_active = false;
  c.aa_target = 0.0;
  if (c.has_people && lane < T) {
    const double* ppl = k.people + (size_t)c.scene * (T + 1) * 6 * N + (size_t)(lane + 1) * 6 * N;
    int closest = -1;
    double best = INFINITY;
    for (int a = 0; a < N; ++a) {
      const double ddx = ppl[a] - c.x0, ddy = ppl[N + a] - c.y0;
      const double d2 = ddx * ddx + ddy * ddy;
      if (d2 < best && ppl[4 * N + a] > 0.05) { best = d2; closest = a; }
    }
    if (closest >= 0 && !(best > 4.0)) {
      const double ax = ppl[closest], ay = ppl[N + closest], ayaw = ppl[2 * N + closest];
      const double agent_angle_initial = atan2(ay - c.y0, ax - c.x0);
      const double hd = ayaw - c.yaw0;
      const double heading_diff = atan2(sin(hd), cos(hd));
      const double rel0 = agent_angle_initial - c.yaw0;
      const double rel = atan2(sin(rel0), cos(rel0));
      const double kThr = M_PI / 6.0, kUp = 5 * M_PI / 6.0;
      if (heading_diff <= -kUp || heading_diff >= kThr) {
        if (!(rel < 0.0)) { c.aa_active = true; c.aa_target = c.yaw0 + (-(M_PI / 6.0)); }
      } else {
        if (!(rel > 0.0)) { c.aa_active = true; c.aa_target = c.yaw0 + (M_PI / 6.0); }
      }
    }
  }
  c.tx = c.ty = 0.0;
  if (lane < T) { c.tx = c.path_pts[2 * (lane + 1)]; c.ty = c.path_pts[2 * (lane + 1) + 1]; }
  c.gx = c.path_pts[2 * T];
  c.gy = c.path_pts[2 * T + 1];
  c.spr = (N > 0) ? kWave / N : 0;
  c.pj = (N > 0) ? lane / N : 0;
  c.pa = (N > 0) ? lane - c.pj * N : 0;
  c.pslot = (N > 0) && (c.pj < c.spr);
  __syncthreads();
}

// ------------------------------------------------------------------------------------------------
// The sweep (kernel K1's body): residuals + Jacobian rows + Gram at parameters x[P] (uniform across lanes).
// Returns the Gram in all lanes; `finite` is false if any residual / Jacobian entry is non-finite.
// When out_r / out_J are non-null (stand-alone K1) the rows are also written to HBM in the reference order.
// ------------------------------------------------------------------------------------------------
template <int NB>
__device__ inline void sweep(Ctx<NB>& c, const double (&x)[2 * NB], Gram<2 * NB>& gram, bool& finite,
                             double* out_r, double* out_J) {
  constexpr int P = 2 * NB;
  const KParams& k = *c.kp;
  const int T = k.T, N = k.N, CH = k.CH, bl = k.bl, lane = c.lane;
  const double dt = k.dt;
  double* pose = c.lds + c.L.pose;
  double* px_ = pose, *py_ = pose + (T + 1), *pth_ = pose + 2 * (T + 1), *pc_ = pose + 3 * (T + 1), *ps_ = pose + 4 * (T + 1);
  const int blast = (CH - 1) / bl;

  // ---- a1 rollout. theta_j: sequential adds in the reference's order; lane j holds theta_j (j = 0..T).
  double th = c.yaw0;
  for (int j = 0; j < T; ++j) {
    const int b = (j < CH) ? j / bl : blast;
    double w = 0.0;
#pragma unroll
    for (int q = 0; q < NB; ++q) w = (q == b) ? x[2 * q + 1] : w;
    if (j < lane) th += w * dt;
  }
  double sn, cs;
  sincos(th, &sn, &cs);
  if (lane <= T) { pth_[lane] = th; pc_[lane] = cs; ps_[lane] = sn; }
  __syncthreads();
  // x, y and sensitivities of pose_{lane+1}: sequential sums over j = 0..lane (reference summation order).
  double X = c.x0, Y = c.y0;
  double Sxv[NB], Syv[NB], Sxw[NB], Syw[NB], Sthw[NB];
#pragma unroll
  for (int q = 0; q < NB; ++q) Sxv[q] = Syv[q] = Sxw[q] = Syw[q] = Sthw[q] = 0.0;
  for (int j = 0; j < T; ++j) {
    const int b = (j < CH) ? j / bl : blast;
    const double cj = pc_[j], sj = ps_[j];
    double v = 0.0;
#pragma unroll
    for (int q = 0; q < NB; ++q) v = (q == b) ? x[2 * q] : v;
    if (j <= lane) {
      X += v * cj * dt;
      Y += v * sj * dt;
      // d x_{j+1} += dt (dv cos - v sin dtheta_j); dtheta_j/dw_q = Sthw[q] (before this step's increment)
#pragma unroll
      for (int q = 0; q < NB; ++q) {
        Sxw[q] = fma(-v * sj * dt, Sthw[q], Sxw[q]);
        Syw[q] = fma(v * cj * dt, Sthw[q], Syw[q]);
        if (q == b) { Sxv[q] += cj * dt; Syv[q] += sj * dt; Sthw[q] += dt; }
      }
    }
  }
  if (lane < T) { px_[lane + 1] = X; py_[lane + 1] = Y; }
  __syncthreads();
  const int myb = (lane < CH) ? lane / bl : blast;  // block driving step `lane`
  double vb = 0.0;
#pragma unroll
  for (int q = 0; q < NB; ++q) vb = (q == myb) ? x[2 * q] : vb;
  // cos/sin of theta_{lane+1} (robot heading at the residual's pose)
  const double c1 = (lane < T) ? pc_[lane + 1] : 1.0, s1 = (lane < T) ? ps_[lane + 1] : 0.0;
  const double th1 = (lane < T) ? pth_[lane + 1] : 0.0;

  // ---- a3 social-work pair rounds
  double soc[kPairOut];
#pragma unroll
  for (int i = 0; i < kPairOut; ++i) soc[i] = 0.0;
  if (c.has_people) {
    double* pair = c.lds + c.L.pair;
    const double* ag = c.lds + c.L.ag;
    const int TN = T * N;
    const int rounds = (T + c.spr - 1) / c.spr;
    for (int r = 0; r < rounds; ++r) {
      const int t = r * c.spr + c.pj;
      double o[kPairOut];
#pragma unroll
      for (int i = 0; i < kPairOut; ++i) o[i] = 0.0;
      if (c.pslot && t < T) {
        const int q = t * N + c.pa;
        const double apx = ag[q], apy = ag[TN + q], awx = ag[2 * TN + q], awy = ag[3 * TN + q];
        const bool valid = ag[4 * TN + q] != 0.0;
        const double rx = px_[t + 1], ry = py_[t + 1], rc = pc_[t + 1], rs = ps_[t + 1];
        const int tb = (t < CH) ? t / bl : blast;
        double rv = 0.0;
#pragma unroll
        for (int qq = 0; qq < NB; ++qq) rv = (qq == tb) ? x[2 * qq] : rv;
        const double rvx = rv * rc, rvy = rv * rs;  // meVel, social_work:170-171
        const double dx = rx - apx, dy = ry - apy;
        const bool degenerate = (dx * dx + dy * dy) < 1e-12;  // sqrt(n2) < 1e-6
        if (valid) {
          // force on the robot from this agent (:125): diff = robot - agent, u = robotVel - agentVel
          const Force F = social_force(dx, dy, rvx - awx, rvy - awy);
          const double dFx_dth = rv * (-rs * F.dfx_dux + rc * F.dfx_duy), dFy_dth = rv * (-rs * F.dfy_dux + rc * F.dfy_duy);
          const double dFx_dv = rc * F.dfx_dux + rs * F.dfx_duy, dFy_dv = rc * F.dfy_dux + rs * F.dfy_duy;
          o[0] = F.fx; o[1] = F.fy;
          o[2] = F.dfx_dx; o[3] = F.dfy_dx; o[4] = F.dfx_dy; o[5] = F.dfy_dy;
          o[6] = dFx_dth; o[7] = dFy_dth; o[8] = dFx_dv; o[9] = dFy_dv;
          if (!degenerate) {
            // force on the agent from the robot (:137-143) is exactly -F for a non-degenerate pair
            o[10] = F.fx * F.fx + F.fy * F.fy;
            o[11] = 2.0 * (F.fx * F.dfx_dx + F.fy * F.dfy_dx);
            o[12] = 2.0 * (F.fx * F.dfx_dy + F.fy * F.dfy_dy);
            o[13] = 2.0 * (F.fx * dFx_dth + F.fy * dFy_dth);
            o[14] = 2.0 * (F.fx * dFx_dv + F.fy * dFy_dv);
          }
        }
        if (!valid || degenerate) {
          // phantom / degenerate: force on the agent from the robot evaluated on its own
          // (diff = agent - robot, u = agentVel - robotVel), derivative signs flip through diff and u.
          const Force G = social_force(-dx, -dy, awx - rvx, awy - rvy);
          const double gx_x = -G.dfx_dx, gy_x = -G.dfy_dx, gx_y = -G.dfx_dy, gy_y = -G.dfy_dy;
          const double gx_th = -rv * (-rs * G.dfx_dux + rc * G.dfx_duy), gy_th = -rv * (-rs * G.dfy_dux + rc * G.dfy_duy);
          const double gx_v = -(rc * G.dfx_dux + rs * G.dfx_duy), gy_v = -(rc * G.dfy_dux + rs * G.dfy_duy);
          o[10] = G.fx * G.fx + G.fy * G.fy;
          o[11] = 2.0 * (G.fx * gx_x + G.fy * gy_x);
          o[12] = 2.0 * (G.fx * gx_y + G.fy * gy_y);
          o[13] = 2.0 * (G.fx * gx_th + G.fy * gy_th);
          o[14] = 2.0 * (G.fx * gx_v + G.fy * gy_v);
        }
      }
#pragma unroll
      for (int i = 0; i < kPairOut; ++i) pair[i * kWave + lane] = o[i];
      __syncthreads();
      // lanes owning the steps of this round sum their N agents in agent order
      const int j = lane - r * c.spr;
      if (j >= 0 && j < c.spr && lane < T) {
        for (int a = 0; a < N; ++a) {
#pragma unroll
          for (int i = 0; i < kPairOut; ++i) soc[i] += pair[i * kWave + j * N + a];
        }
      }
      __syncthreads();
    }
  }

  // ---- per-step rows (lane t < T): state-space gradients (gx, gy, gth) + direct dv on block myb
  gram.clear();
  finite = true;
  const smpc_params& w = k.prm;
  const int rows_per_step = c.has_people ? 8 : 5;
  const int row0 = rows_per_step * lane + min(max(lane - 1, 0), k.nfeas) + ((lane >= 1 && lane - 1 < k.nfeas) ? 0 : 0);
  int rowi = row0;
  auto emit = [&](double r, double gx, double gy, double gth, double gv) {
    double row[P];
#pragma unroll
    for (int q = 0; q < NB; ++q) {
      row[2 * q] = gx * Sxv[q] + gy * Syv[q] + ((q == myb) ? gv : 0.0);
      row[2 * q + 1] = gx * Sxw[q] + gy * Syw[q] + gth * Sthw[q];
    }
    bool ok = isfinite(r);
#pragma unroll
    for (int q = 0; q < P; ++q) ok = ok && isfinite(row[q]);
    if (!ok) finite = false;
    gram.add_row(row, r);
    if (out_r) out_r[rowi] = r;
    if (out_J) {
#pragma unroll
      for (int q = 0; q < P; ++q) out_J[(size_t)rowi * P + q] = row[q];
    }
    ++rowi;
  };
  if (lane < T) {
    if (c.has_people) {
      // a7 agent angle
      {
        double r = 0.0, gth = 0.0;
        if (c.aa_active) {
          const double u = th1 - c.aa_target;
          const double ad = atan2(sin(u), cos(u));
          r = w.agent_angle_w * (ad * ad);
          gth = w.agent_angle_w * 2.0 * ad;
        }
        emit(r, 0.0, 0.0, gth, 0.0);
      }
      // a3 social work: w (|sum F|^2 + sum |G|^2 + 1e-6)
      {
        const double wr = soc[0] * soc[0] + soc[1] * soc[1];
        const double r = w.socialwork_w * (wr + soc[10] + 1e-6);
        const double gx = w.socialwork_w * (2.0 * (soc[0] * soc[2] + soc[1] * soc[3]) + soc[11]);
        const double gy = w.socialwork_w * (2.0 * (soc[0] * soc[4] + soc[1] * soc[5]) + soc[12]);
        const double gt = w.socialwork_w * (2.0 * (soc[0] * soc[6] + soc[1] * soc[7]) + soc[13]);
        const double gv = w.socialwork_w * (2.0 * (soc[0] * soc[8] + soc[1] * soc[9]) + soc[14]);
        emit(r, gx, gy, gt, gv);
      }
      // a4 proxemics: w alpha exp(-min_a d^2 / d0^2) over valid agents (first minimum wins)
      {
        const double* ag = c.lds + c.L.ag;
        const int TN = T * N;
        double best = 1.7976931348623157e308, bdx = 0.0, bdy = 0.0;
        for (int a = 0; a < N; ++a) {
          const int q = lane * N + a;
          if (ag[4 * TN + q] == 0.0) continue;
          const double ddx = X - ag[q], ddy = Y - ag[TN + q];
          const double d2 = ddx * ddx + ddy * ddy;
          if (d2 < best) { best = d2; bdx = ddx; bdy = ddy; }
        }
        const double e = 3.0 * exp(-best / (0.5 * 0.5));
        const double r = w.proxemics_w * e;
        const double gx = r * (-2.0 * bdx / (0.5 * 0.5)), gy = r * (-2.0 * bdy / (0.5 * 0.5));
        emit(r, gx, gy, 0.0, 0.0);
      }
    }
    // a6 velocity
    {
      double r = 0.0, gv = 0.0;
      if (lane < CH) { const double d = w.desired_linear_vel - vb; r = w.velocity_w * d * d; gv = -2.0 * w.velocity_w * d; }
      emit(r, 0.0, 0.0, 0.0, gv);
    }
    // a8 goal align
    {
      const double u = c.goal_yaw - th1;
      const double a = atan2(sin(u), cos(u));
      emit(w.goal_align_w * a * a, 0.0, 0.0, -2.0 * w.goal_align_w * a, 0.0);
    }
    // a2 distance (path follow -> final point; path align -> point t+1)
    {
      const double ddx = X - c.gx, ddy = Y - c.gy, q2 = ddx * ddx + ddy * ddy;
      emit(w.distance_w * q2 * q2, 4.0 * w.distance_w * q2 * ddx, 4.0 * w.distance_w * q2 * ddy, 0.0, 0.0);
    }
    {
      const double ddx = X - c.tx, ddy = Y - c.ty, q2 = ddx * ddx + ddy * ddy;
      emit(w.angle_w * q2 * q2, 4.0 * w.angle_w * q2 * ddx, 4.0 * w.angle_w * q2 * ddy, 0.0, 0.0);
    }
    // a5 obstacle
    {
      const double fxp = X + 0.25 * c1, fyp = Y + 0.25 * s1;
      const double inv_res = 1.0 / k.resolution;
      const double ic = (fxp - c.ox) / k.resolution, ir = (fyp - c.oy) / k.resolution;
      double f, dfdr, dfdc;
      bicubic(c.map, k.size_x, k.size_y, ir, ic, f, dfdr, dfdc);
      const double gx = w.obstacle_w * dfdc * inv_res, gy = w.obstacle_w * dfdr * inv_res;
      const double gth = w.obstacle_w * (dfdc * (-0.25 * s1) + dfdr * (0.25 * c1)) * inv_res;
      emit(w.obstacle_w * f, gx, gy, gth, 0.0);
    }
    // a9 velocity feasibility between blocks `lane` and `lane-1` (src/optimizer.cpp:364-370); row follows step `lane`
    if (lane >= 1 && lane <= k.nfeas) {
      double row[P];
#pragma unroll
      for (int q = 0; q < P; ++q) row[q] = 0.0;
      double lin = 0.0, ang = 0.0;
#pragma unroll
      for (int q = 1; q < NB; ++q) {
        if (q == lane) {
          lin = x[2 * q] - x[2 * q - 2];
          ang = x[2 * q + 1] - x[2 * q - 1];
        }
      }
      double r = 0.0;
      if (lane < CH) {
        r = w.velocity_feasibility_w * lin * lin + w.velocity_feasibility_w * ang * ang;
#pragma unroll
        for (int q = 1; q < NB; ++q) {
          if (q == lane) {
            row[2 * q] = 2.0 * w.velocity_feasibility_w * lin;
            row[2 * q - 2] = -2.0 * w.velocity_feasibility_w * lin;
            row[2 * q + 1] = 2.0 * w.velocity_feasibility_w * ang;
            row[2 * q - 1] = -2.0 * w.velocity_feasibility_w * ang;
          }
        }
      }
      if (!isfinite(r)) finite = false;
      gram.add_row(row, r);
      if (out_r) out_r[rowi] = r;
      if (out_J) {
#pragma unroll
        for (int q = 0; q < P; ++q) out_J[(size_t)rowi * P + q] = row[q];
      }
      ++rowi;
    }
  }
  // ---- Gram reduction over lanes (all lanes end with the sums)
#pragma unroll
  for (int i = 0; i < Gram<P>::SZ; ++i) gram.v[i] = wave_sum(gram.v[i]);
  finite = !wave_any(!finite);
}

}  // namespace smpc
