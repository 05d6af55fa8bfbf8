// smpc_device.hpp — gfx950 device code of the batched social-MPC solver: the residual/Jacobian sweep (K1).
//
// What this restates, MI355X-first (reference files relative to /root/reference):
//   * rollout a1 (include/nav2_social_mpc_controller/update_state.hpp:37-63) as ONE shared rollout per sweep
//     plus closed-form sensitivities S_t = d(x,y,theta)_{t+1}/d(params), instead of the reference's per-functor
//     O(t) re-integration on ceres::Jet;
//   * the nine instantiated residual kinds a2..a9 (include/.../critics/*_cost_function.hpp) with analytic
//     state-space gradients (x, y, theta, v_block) chained with S_t — equal to Ceres autodiff in exact arithmetic;
//   * the Gram contraction [J r]^T [J r] (gives J^T J, J^T r and the cost in one pass).
//
// Lane mapping: a 64-lane wavefront is split into S = 64/W "slots" of W lanes (W = 32 when T+1 <= 32, else 64);
// each slot works on its own scene, lane `sl` of a slot owns horizon step t = sl (pose after t+1 steps) for every
// critic, and walks the N agents of that step in a register-resident loop (no cross-lane traffic for the social
// terms). The horizon's cos/sin block and the staged people block live in LDS; per-step reductions over the slot
// use wavefront shuffles.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/smpc.h"
#include "smpc_math.hpp"

namespace smpc {

constexpr int kWave = 64;
constexpr double kNoTarget = 1e300;  // agent-angle tag: no steering target at this step
constexpr int kSoc = 15;  // sum F(2), sum dF/d{x,y,th,v}(8), sum |G|^2 (1), sum d|G|^2/d{x,y,th,v} (4)

struct KParams {
  int B, T, N, CH, bl, nb, P, nbounded, nfeas;
  int size_x, size_y, costmap_shared;
  double dt, resolution;
  double inv_resolution;  // 1 / resolution, rounded once on the host
  smpc_params prm;
  const double* pose0;
  const double* init_params;
  const double* path_pts;
  const double* goal_yaw;
  const double* people;
  const uint8_t* has_people;
  const int32_t* T_scene;  // [B] rollout steps of each scene (<= T), or null: every scene has T (smpc_scene_batch.T_scene)
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
  int* queue;  // scene work queue (one int, zeroed before every solve launch)
  // staged people block (smpc_stage_people_batch / the library's own staging pass): what the sweep reads
  const double* people_rec;  // [B][N][T][4]  px, py, vx, vy of people_proj[t + 1][a]
  const double* people_aux;  // [B][T][2]     bit mask of valid agents (u64 bits), agent-angle target (kNoTarget: none)
  const int32_t* order;      // [B] queue order of the solve kernel (null: index order)
  int hp_A;                  // helper lanes (W = 64 kernels, see sweep()): agents the owner lane of a step walks itself;
                             // == N: no helpers. Agents hp_A .. N-1 of every step are walked by the lanes beyond the horizon
  int full_gram;             // != 0: every sweep of a solve forms the whole Gram (SMPC_FULL_GRAM: the check that stopping at
                             // its last column changes nothing)
  int prio_step;             // > 0: a wave whose oldest scene has made n sweeps runs at wave priority min(n / prio_step, 3)
  double* stage_rec;         // staging kernel outputs (same layouts)
  double* stage_aux;
  unsigned long long* stamps;  // diagnostic builds only (SMPC_STAMPS): per-wave cycle sums per phase, [grid][8]
  // eval (K1) inputs / outputs
  const double* e_x;
  double* e_residuals;
  double* e_jacobian;
  double* e_cost;
  double* e_gradient;
  int e_M;  // row stride of the eval outputs (M with people)
  int e_row_order;  // 0: reference (step-major) row order, 1: critic-major (smpc_eval_batch_out.row_order)
  MathTab mt;  // polynomial coefficients of smpc_math.hpp, read through scalar loads
  AtanNodeTab an;  // nodes of atan2_unit(), copied into LDS by every wave (load_atan_nodes)
};

// Cross-lane sum of the per-lane Gram shares (solve kernel): values go through LDS in chunks of whole columns of the
// packed upper triangle, at most kGramChunk values at a time; lane (part, v) of a slot then adds up value v of the 16
// lanes of its part and the parts are combined by shuffles (W / 16 + 3 additions instead of 3 log2(W) shuffle
// instructions per value). The buffer (W rows of kGramChunk + 1 doubles) lies over the sweep's own temporaries — the
// cos / sin block and the scans are dead once the sensitivities are formed — plus a tail of its own; outside the
// sweep the same area holds the temporaries of the LM algebra.
constexpr int kSensInRegsMaxBlocks = 6;  // K1 keeps a lane's sensitivities in registers up to this many parameter blocks
// doubles of wave-shared LDS of the K1 kernel: two row staging blocks per slot, and the parked sensitivities
__host__ __device__ constexpr int eval_extra_doubles(int T, int P, int W) {
  return (64 / W) * (2 * T * P) + (P / 2 > kSensInRegsMaxBlocks ? (64 / W) * (5 * (P / 2) * W) : 0);
}
// The node table of atan2_unit() sits behind everything else in a wave's LDS (solve and K1 kernels).
__host__ __device__ constexpr int atan_tab_offset(int slot_doubles, int extra_doubles) { return (slot_doubles + extra_doubles + 3) & ~3; }
constexpr int kAtanTabDoubles = kAtanNodes * kAtanNodeStride;
constexpr int kGramChunk = 16;
__host__ __device__ constexpr int gram_red_doubles(int W) { return W * (kGramChunk + 1); }
// doubles of wave-shared LDS behind the per-slot blocks of the solve kernel: the feasibility rows of every slot
__host__ __device__ constexpr int wave_extra_doubles(int P, int W) {
  return (kWave / W) * ((P / 2 > 1 ? P / 2 - 1 : 1) * (P + 1));
}

// LDS carve-up (in doubles) of ONE slot.
struct LdsLayout {
  int ag;       // [N][T][4]  staged people block (px, py, vx, vy) — staging kernel only; the sweep reads the staged
                //            records from global memory (HBM once in K1, L2 on the later sweeps of a solve)
  int valid;    // [T]        bit a set = agent a valid at step t (64-bit words)
  int cs;       // [2][T+1]   cos, sin of theta_j, j = 0..T
  int inc;      // [4][T+1]   inclusive scans over j of cos, sin, j cos, j sin(theta_j) of the current sweep
  int cst;      // [8]        x0, y0, yaw0, goal_yaw, origin x, origin y, final point x, y
  int stepst;   // [4][T]     helper lanes only: robot x, y, velocity x, y at every step (what a pair evaluation needs of it)
  int part;     // [T][kPart] helper lanes only: the partial sums a helper hands to the owner lane of a step
  int hz;       // [4]        the scene's own horizon (kernels with per-scene T): ints T, CH, bl, last block, feasibility
                //            rows, bounded blocks
  int lanec;    // [3][T]     per step: path point x, y (path_pts[t+1]) and agent-angle target (kNoTarget = none)
  int lm;       // LM vectors / matrices / scalars
  int gram;     // [(P+1)^2] Gram [J r]^T [J r] of the latest sweep, dense and symmetric
  int scratch;  // polynomial scratch
  int total;
};

enum LayoutKind { kLayoutEval = 0, kLayoutSolve = 1, kLayoutStage = 2 };

// kLayoutSolve: LM state in LDS. kLayoutEval: the stand-alone K1 kernel, a single sweep. kLayoutStage: the staging
// kernel, people block in LDS on its way to the staged records.
constexpr int kPart = 18;  // 6 + 4 + 1 + 4 partial sums, nearest distance, its agent index, redo flag (+ 1 spare)

// Helper lanes. With one scene per wave (W = 64) the lanes beyond the horizon (64 - T of them) idle through the agent loop,
// the longest part of a sweep. They take over the tail of every step's agent list instead: the owner lane of step t walks
// agents 0 .. A-1, one helper walks agents A .. N-1 of step t (a "unit"), helper h taking the units h, h + R, h + 2R, ...
// (R = 64 - T helpers, U = ceil(T / R) units each) and handing the partial sums of each unit to its owner through LDS.
// A is the smallest count with U (N - A) <= A: owners and helpers then finish together after A iterations instead of N
// (BASELINE configs[4]: N = 16, T = 38 -> A = 11; params.yaml shape N = 3 -> A = 2). Returns N when helpers do not
// pay (each unit costs a flush of ~30 instructions, the hand-over another ~40 per step).
__host__ __device__ inline int helper_owner_agents(int T, int N, int W) {
  const int R = W - T;
  if (W != 64 || R < 1 || N < 2) return N;
  const int U = (T + R - 1) / R;
  const int A = (U * N + U) / (U + 1);            // ceil(U N / (U + 1))
  if (A >= N) return N;
  // instructions saved per sweep against the hand-over's, with a margin of two: measured, the params.yaml shape (N = 3:
  // one pair saved, two units flushed) gained nothing, BASELINE configs[4] (five pairs saved) 14 %
  return ((N - A) * 250 > 2 * (60 * U + 120)) ? A : N;
}

// W: the slot width of the kernel the layout is for (32: two scenes per wave, 64: one; slot_width() / solve_slot_width())
__host__ __device__ inline LdsLayout make_layout(int T, int N, int P, int kind, int W) {
  LdsLayout L;
  const bool with_lm = kind == kLayoutSolve;
  int o = 0;
  L.ag = o; if (kind == kLayoutStage) o += 4 * T * (N > 0 ? N : 1);
  L.valid = o; o += T;
  L.cs = o; o += 2 * (T + 1);
  L.inc = o; o += 4 * (T + 1);
  if (with_lm) {  // tail of the Gram reduction buffer / LM temporaries, which start at L.cs
    const int want = gram_red_doubles(W) > P * P + 7 * P + 96 ? gram_red_doubles(W) : P * P + 7 * P + 96;
    if (want > 6 * (T + 1)) o += want - 6 * (T + 1);
  }
  L.cst = o; o += 8;
  L.hz = o; o += 4;
  L.stepst = o; L.part = o;
  if (kind != kLayoutStage && helper_owner_agents(T, N, W) < N) { o += 4 * T; L.part = o; o += kPart * T; }
  L.lanec = o; o += 3 * T;
  L.lm = o; if (with_lm) o += P * P + 6 * P + 24;  // Hs, six vectors, scalars: what lives from trip to trip
  L.gram = o; o += (P + 1) * (P + 1);  // dense symmetric [J r]^T [J r] of the latest sweep (VALU back-end)
  L.scratch = o;  // (the generic line-search interpolation fallback borrows the wave's Gram reduction buffer)
  L.total = (o + 3) & ~3;  // 32-byte multiple: records are moved as 4-double vectors
  return L;
}

__host__ __device__ inline int slot_width(int T, int N) { return (T + 1 <= 32 && N <= 32) ? 32 : 64; }

// The workgroup is ONE wavefront: LDS operations of a wave execute in program order, so cross-lane hand-offs through
// LDS only need the compiler not to reorder them — no s_barrier and, importantly, no s_waitcnt vmcnt(0) that a
// __syncthreads() would add (it would drain unrelated global loads / stores at every hand-off).
__device__ inline void wave_lds_fence() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <int W> __device__ inline double slot_sum(double v) {
#pragma unroll
  for (int off = W / 2; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

__device__ inline double wrap_to_pi(double a) {  // critics/social_work_cost_function.hpp:39-46
  // only ever called on a difference of two atan2 results (|a| <= 2 pi, or NaN): at most one trip per loop; the guard
  // keeps the wave finite should that ever change
  if (!(fabs(a) <= 8.0 * M_PI)) a = fmod(a, 2.0 * M_PI);
  while (a > M_PI) a -= 2.0 * M_PI;
  while (a <= -M_PI) a += 2.0 * M_PI;
  return a;
}

// atan2(sin u, cos u) restated as a range reduction of u into (-pi, pi]; equal up to round-off
// (critics/agent_angle_cost_function.hpp:156, critics/goal_align_cost_function.hpp:111-112).
__device__ inline double wrap_angle(double u) {
  const double k = rint(u * (0.5 / M_PI));
  double r = fma(-k, 2.0 * M_PI, u);
  r = fma(-k, 2.4492935982947064e-16, r);  // 2*pi - (double)(2*pi)
  return r;
}

// ------------------------------------------------------------------------------------------------
// Social force between one (me, other) pair and its derivatives with respect to diff = me_pos - other_pos and
// u = me_vel - other_vel. Restates computeSocialForce (critics/social_work_cost_function.hpp:164-228) for a
// single "other"; constants from src/critics/social_work_cost_function.cpp:38-43. F = k (fv i + fa i_perp).
// ------------------------------------------------------------------------------------------------
struct Force {
  double fx, fy;
  double dfx_dx, dfy_dx, dfx_dy, dfy_dy;      // wrt diff
  double dfx_dux, dfy_dux, dfx_duy, dfy_duy;  // wrt u
  bool special;  // pair_force() only: this pair needs social_force_general() (theta within 1e-6 of 0 or pi)
};

// 1/sqrt(x) for a normal positive x: hardware estimate + one cubic correction (what the library routine does, without
// its zero / infinity / denormal cases, which cannot occur here: d2 >= 1e-12 after the coincident-pair clamp, and a
// vanishing interaction vector is a singular configuration for the reference as well).
__device__ inline double fast_rsqrt(double x) {
  const double y = __builtin_amdgcn_rsq(x);
  const double e = fma(-x * y, y, 1.0);
  return fma(y * e, fma(e, 0.375, 0.5), y);
}

__device__ inline Force social_force_general(double dx, double dy, double ux, double uy) {
  const double lambda = 2.0, gamma = 0.35, nPrime = 3.0, nn = 2.0, k = 2.1;
  Force R;
  double d2 = dx * dx + dy * dy;
  const bool degenerate = d2 < 1e-12;  // |diff| < 1e-6 (:181-184): diff := (1e-6, 0), a constant: no dependence on positions
  if (degenerate) { dx = 1e-6; dy = 0.0; d2 = 1e-12; }
  const double inv_n = fast_rsqrt(d2);
  const double n = d2 * inv_n;
  const double ex = dx * inv_n, ey = dy * inv_n;  // diffDirection :185
  const double ivx = fma(lambda, ux, ex), ivy = fma(lambda, uy, ey);  // :191-192 (lambda u is exact: lambda = 2)
  const double L2 = ivx * ivx + ivy * ivy;
  const double inv_L = fast_rsqrt(L2);
  const double L = L2 * inv_L;  // :194
  const double ix = ivx * inv_L, iy = ivy * inv_L;  // :195-196
  // theta = wrapToPi(atan2(dir) - atan2(idir)) (:198-200) is the angle from idir to dir = atan2(idir x dir, idir . dir).
  // One atan2 instead of two wherever that cannot change sign(theta): away from theta = 0 and |theta| = pi
  // (|sin theta| >= 1e-6). Closer than that the reference's own two-atan2 form is evaluated, so that the last-bit
  // behaviour next to the discontinuity of sign(theta) (:210) stays the reference's.
  // Equal velocities (robot stopped beside a standing person): theta is mathematically 0 and the reference's value is
  // 0 up to the last-bit noise of its own libm (which then decides sign(theta)). Take exactly 0.
  const double cross = ix * ey - iy * ex, dot = ix * ex + iy * ey;
  double phi;
  if (ux == 0.0 && uy == 0.0) {
    phi = 0.0;
  } else if (fabs(cross) >= 1e-6) {
    phi = atan2(cross, dot);
  } else {
    phi = wrap_to_pi(atan2(ey, ex) - atan2(iy, ix));
  }
  const double Bq = gamma * L;  // :203
  const double inv_B = inv_L * (1.0 / gamma);
  const double a1 = nPrime * Bq * phi, a2 = nn * Bq * phi;
  const double base = -n * inv_B;
  const double E1 = exp(base - a1 * a1);  // :205-207
  const double E2 = exp(base - a2 * a2);  // :212-215
  const double sgn = (phi > 0.0) ? 1.0 : -1.0;  // :210
  const double fv = -E1, fa = -sgn * E2;
  R.fx = k * (fv * ix - fa * iy);  // :218-224, i_perp = (-iy, ix)
  R.fy = k * (fv * iy + fa * ix);
  // The force depends on its inputs through (n, alpha = atan2(e), iv): dF = A_n dn + A_alpha dalpha + A_x d(iv_x) +
  // A_y d(iv_y). With dL = i . d(iv), kappa = d atan2(i) = (i_perp . d(iv)) / L, dphi = dalpha - kappa, dB = gamma dL,
  //   d(arg_m) = -dn/B + n dB/B^2 - 2 a_m c_m (dB phi + B dphi),   dF = k ((dfv - fa kappa) i + (dfa + fv kappa) i_perp),
  // the four columns are evaluated once and every direction below is a linear combination of them.
  const double nB2 = n * inv_B * inv_B;
  // A_n: dn = 1 -> d(arg) = -1/B, no rotation: -F / B
  const double anx = -inv_B * R.fx, any = -inv_B * R.fy;
  // A_alpha: dalpha = 1 -> dphi = 1, d(arg_m) = -2 a_m c_m B
  double aax, aay;
  {
    const double twoB = 2.0 * Bq;
    const double dfv = E1 * (a1 * nPrime * twoB);
    const double dfa = sgn * E2 * (a2 * nn * twoB);
    aax = k * (dfv * ix - dfa * iy);
    aay = k * (dfv * iy + dfa * ix);
  }
  // A_x, A_y: d(iv) = (1, 0) / (0, 1)
  auto column = [&](double dL, double kappa, double& ofx, double& ofy) {
    const double dB = gamma * dL;
    const double dbase = nB2 * dB;
    const double common = dB * phi - Bq * kappa;  // dphi = -kappa
    const double dfv = -E1 * (dbase - 2.0 * a1 * nPrime * common);
    const double dfa = -sgn * E2 * (dbase - 2.0 * a2 * nn * common);
    const double ci = dfv - fa * kappa, cp = dfa + fv * kappa;
    ofx = k * (ci * ix - cp * iy);
    ofy = k * (ci * iy + cp * ix);
  };
  double axx, axy, ayx, ayy;
  column(ix, -iy * inv_L, axx, axy);
  column(iy, ix * inv_L, ayx, ayy);
  R.dfx_dux = lambda * axx; R.dfy_dux = lambda * axy;  // u enters iv as lambda u
  R.dfx_duy = lambda * ayx; R.dfy_duy = lambda * ayy;
  if (degenerate) {
    R.dfx_dx = R.dfy_dx = R.dfx_dy = R.dfy_dy = 0.0;
  } else {
    // moving diff by dd: dn = e . dd, dalpha = (e_perp . dd) / n, d(iv) = de = e_perp dalpha, e_perp = (-ey, ex);
    // C = A_alpha + A_x (-ey) + A_y ex is what one unit of dalpha does in total
    const double cx = aax - ey * axx + ex * ayx, cy = aay - ey * axy + ex * ayy;
    const double da1 = -ey * inv_n, da2 = ex * inv_n;  // dd = (1, 0) / (0, 1)
    R.dfx_dx = ex * anx + da1 * cx; R.dfy_dx = ex * any + da1 * cy;
    R.dfx_dy = ey * anx + da2 * cx; R.dfy_dy = ey * any + da2 * cy;
  }
  return R;
}

// The same force for a regular pair (|diff| >= 1e-6, the overwhelmingly common case), built for instruction count:
// table-driven exp / atan2 (smpc_math.hpp), no selects for the coincident-pair clamp, no branches. Two rare shapes
// are only flagged, for the caller to redo the step's agents with social_force_general(): a coincident pair (flagged
// by the caller) and a pair whose theta is within 1e-6 of 0 or pi while the velocities differ (Force::special: next
// to the discontinuity of sign(theta) the reference's own two-atan2 form decides, :198-200).
// pair_force(-d, -u) == -pair_force(d, u) bit for bit (every intermediate flips sign or stays exactly), with equal
// derivatives: the force on an agent from the robot needs no evaluation of its own.
// The constant factors are left to the caller, who applies them once to the sums over the agents of a step instead of
// to every pair: the force and its diff-derivatives come WITHOUT the factor k (kPairForceK), the u-derivatives without
// k * lambda (kPairForceLambda; u enters the interaction vector as lambda u).
constexpr double kPairForceK = 2.1, kPairForceLambda = 2.0;
__device__ inline Force pair_force(MathTabP mt, const double* atab, double dx, double dy, double ux, double uy) {
  const double lambda = kPairForceLambda, gamma = 0.35, nPrime = 3.0, nn = 2.0;
  Force R;
  const double d2 = fma(dx, dx, dy * dy);
  const double inv_n = rsqrt_pos(d2);
  const double n = d2 * inv_n;
  const double ex = dx * inv_n, ey = dy * inv_n;  // diffDirection :185
  const double ivx = fma(lambda, ux, ex), ivy = fma(lambda, uy, ey);  // :191-192
  const double L2 = fma(ivx, ivx, ivy * ivy);
  const double inv_L = rsqrt_pos(L2);
  const double L = L2 * inv_L;  // :194
  const double ix = ivx * inv_L, iy = ivy * inv_L;  // :195-196
  const double cross = fma(ix, ey, -(iy * ex)), dot = fma(ix, ex, iy * ey);
  const bool zero_u = (ux == 0.0) & (uy == 0.0);  // equal velocities: theta := 0 (parity)
  double phi = atan2_unit(mt, atab, cross, dot);  // (cross, dot) = (sin, cos) of theta: a unit vector
  // keep the scheduler from interleaving the arctangent, the two exponentials and the derivative block: the extra
  // overlap buys nothing with two or three waves per SIMD and costs ~15 VGPRs (the stand-alone K1 kernel would drop
  // from three waves per SIMD to two)
  __builtin_amdgcn_sched_barrier(0);
  R.special = !zero_u & (fabs(cross) < 1e-6);
  phi = zero_u ? 0.0 : phi;
  const double Bq = gamma * L;  // :203
  const double inv_B = inv_L * (1.0 / gamma);
  const double a1 = nPrime * Bq * phi, a2 = nn * Bq * phi;
  const double base = -n * inv_B;
  const double E1 = exp_tab(mt, fma(-a1, a1, base));  // :205-207
  const double E2 = exp_tab(mt, fma(-a2, a2, base));  // :212-215
  __builtin_amdgcn_sched_barrier(0);
  const double fv = -E1;
  const double fa = (phi > 0.0) ? -E2 : E2;  // -sign(theta) E2, sign = -1 at theta == 0 (:210)
  R.fx = fma(fv, ix, -(fa * iy));  // :218-224 without the factor k, i_perp = (-iy, ix)
  R.fy = fma(fv, iy, fa * ix);
  // derivative: see social_force_general(); dfa = sgn E2 (...) = -fa (...)
  const double nB2 = n * inv_B * inv_B;
  const double anx = -inv_B * R.fx, any = -inv_B * R.fy;
  const double twoB = 2.0 * Bq;
  const double g1 = a1 * nPrime, g2 = a2 * nn;
  double aax, aay;
  {
    const double dfv = E1 * (g1 * twoB);
    const double dfa = -fa * (g2 * twoB);
    aax = fma(dfv, ix, -(dfa * iy));
    aay = fma(dfv, iy, dfa * ix);
  }
  auto column = [&](double dL, double kappa, double& ofx, double& ofy) {
    const double dB = gamma * dL;
    const double dbase = nB2 * dB;
    const double common = 2.0 * fma(dB, phi, -(Bq * kappa));  // dphi = -kappa
    const double dfv = fv * fma(-g1, common, dbase);
    const double dfa = fa * fma(-g2, common, dbase);
    const double ci = fma(-fa, kappa, dfv), cp = fma(fv, kappa, dfa);
    ofx = fma(ci, ix, -(cp * iy));
    ofy = fma(ci, iy, cp * ix);
  };
  double axx, axy, ayx, ayy;
  column(ix, -iy * inv_L, axx, axy);
  column(iy, ix * inv_L, ayx, ayy);
  R.dfx_dux = axx; R.dfy_dux = axy;  // without the factor k * lambda
  R.dfx_duy = ayx; R.dfy_duy = ayy;
  const double cx = aax - ey * axx + ex * ayx, cy = aay - ey * axy + ex * ayy;
  const double da1 = -ey * inv_n, da2 = ex * inv_n;
  R.dfx_dx = ex * anx + da1 * cx; R.dfy_dx = ex * any + da1 * cy;
  R.dfx_dy = ey * anx + da2 * cx; R.dfy_dy = ey * any + da2 * cy;
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

// The same interpolation split in two, so that the sweep can request the 4 x 4 patch as soon as the pose is known and
// consume it after the agent loop (the 16 dependent byte loads were 6 % of a lone solve launch): the patch is fetched
// as four unaligned dwords, one per row, starting at column clamp(col - 1, 0, size_x - 4); the byte of clamped column
// cc is then byte (cc - start) of its row's dword — also at the edges, where several taps share a cell.
struct CostPatch {
  uint32_t row[4];   // bytes start .. start + 3 of the four clamped rows
  uint32_t sh;       // bit offsets (0, 8, 16, 24) of the four taps inside a row dword, 5 bits each
};

// Integer cell of a coordinate, kept defined for wild values (clamping below makes any far-outside index equivalent).
__device__ inline int cell_index(double v, int size) { return (int)fmin(fmax(floor(v), -4.0), (double)size + 4.0); }

// true when the whole 4 x 4 patch around (r, c) lies inside the map: no tap is clamped (NaN coordinates: false)
__device__ inline bool bicubic_interior(int size_x, int size_y, double r, double c) {
  const int row = cell_index(r, size_y), col = cell_index(c, size_x);
  return (row >= 1) & (row <= size_y - 3) & (col >= 1) & (col <= size_x - 3);
}

// kInterior: the caller has established bicubic_interior() for EVERY lane of the wave (a wave-uniform decision: the
// clamps, the per-tap bit offsets and their variable shifts — ~90 integer instructions per sweep — are then skipped;
// the taps and the arithmetic on them are the same, so the result does not depend on which path a wave takes).
template <bool kInterior>
__device__ inline void bicubic_fetch(const uint8_t* __restrict__ map, int size_x, int size_y, double r, double c, CostPatch& p) {
  const int row = cell_index(r, size_y), col = cell_index(c, size_x);
  if (kInterior) {
    const uint8_t* q = map + (size_t)(row - 1) * size_x + (col - 1);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      uint32_t v;
      __builtin_memcpy(&v, q + (size_t)i * size_x, 4);  // unaligned dword
      p.row[i] = v;
    }
    p.sh = 0u | (8u << 5) | (16u << 10) | (24u << 15);
    return;
  }
  const int start = min(max(col - 1, 0), size_x - 4);
  uint32_t sh = 0;
#pragma unroll
  for (int j = 0; j < 4; ++j) sh |= (uint32_t)(8 * (min(max(col - 1 + j, 0), size_x - 1) - start)) << (5 * j);
  p.sh = sh;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int rr = min(max(row - 1 + i, 0), size_y - 1);
    uint32_t v;
    __builtin_memcpy(&v, map + (size_t)rr * size_x + start, 4);  // unaligned dword
    p.row[i] = v;
  }
}

// (r, c) must be the coordinates the patch was fetched for
template <bool kInterior>
__device__ inline void bicubic_eval(const CostPatch& p, double r, double c, double& f, double& dfdr, double& dfdc) {
  const double tr = r - floor(r), tc = c - floor(c);
  double fv[4], dv[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    double t[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (kInterior) t[j] = (double)((p.row[i] >> (8 * j)) & 0xffu);
      else t[j] = (double)((p.row[i] >> ((p.sh >> (5 * j)) & 31u)) & 0xffu);
    }
    cubic_hermite(t[0], t[1], t[2], t[3], tc, fv[i], dv[i]);
  }
  double unused;
  cubic_hermite(fv[0], fv[1], fv[2], fv[3], tr, f, dfdr);
  cubic_hermite(dv[0], dv[1], dv[2], dv[3], tr, dfdc, unused);
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
// Per-slot scene context (registers; slot-uniform unless noted)
// ------------------------------------------------------------------------------------------------
// In-kernel phase stamps (diagnostic build -DSMPC_STAMPS only; the shipped kernel executes none of this).
#ifdef SMPC_STAMPS
#define SMPC_STAMP(ctx, phase) do { const unsigned long long _t = __builtin_amdgcn_s_memtime(); (ctx).acc[phase] += _t - (ctx).t_last; (ctx).t_last = _t; } while (0)
#define SMPC_STAMP2(ctx, phase) do { const unsigned long long _t = __builtin_amdgcn_s_memtime(); (ctx).acc2[phase] += _t - (ctx).t_last; (ctx).acc[5] += _t - (ctx).t_last; (ctx).t_last = _t; } while (0)
#else
#define SMPC_STAMP2(ctx, phase) do { } while (0)
#define SMPC_STAMP(ctx, phase) do { } while (0)
#endif

// Kernel parameters are read through the kernel-argument segment (constant address space) instead of being held in
// SGPRs for the whole kernel: the ~90 scalars of KParams otherwise overflow the SGPR file and come back as
// v_readlane / v_writelane spill traffic on the VALU (1.4 k such instructions in the solve kernel before).
typedef const KParams __attribute__((address_space(4))) * KParamsK;

struct Ctx {
#ifdef SMPC_STAMPS
  unsigned long long t_last;
  unsigned long long acc[8];
  unsigned long long acc2[4];
#endif
  KParamsK kp;  // the launch parameters, read where they lie in the kernel-argument segment
  int scene;   // scene index of this slot
  int sl;      // lane within the slot = horizon step owned by this lane (per-lane)
  int slot;
  bool has_people;
  bool gram_full;  // solve kernel: the latest sweep left the whole Gram in LDS, not only its last column (wave-uniform)
  const uint8_t* map;
  double* lds;       // this slot's LDS block (scene constants, cos/sin block, LM state live here)
  const double* ag;  // staged people records [N][T][4] of this slot's scene (global memory)
  double* wave_lds;  // wave-shared LDS behind the slot blocks (solve: feasibility rows of every slot; K1: row staging blocks)
  const double* atab;  // the wave's copy of the atan2_unit() nodes (LDS)
  LdsLayout L;
};

// every lane of the wave; the caller fences before the first sweep
__device__ inline void load_atan_nodes(KParamsK kp, double* dst, int lane) {
  for (int i = lane; i < kAtanTabDoubles; i += kWave) dst[i] = kp->an.v[i];
}

// Dense symmetric view of the slot's Gram [J r]^T [J r] left in LDS by sweep(): G(a, b), a, b in 0..P (column P = r).
struct GramView {
  const double* base;
  int ld;
  __device__ inline double operator()(int a, int b) const { return base[a * ld + b]; }
};

typedef double v4d __attribute__((ext_vector_type(4)));
typedef double v2d __attribute__((ext_vector_type(2)));

// Staging pass (its own kernel, once per people block): gather the slot's people block, agent index fastest across
// lanes (coalesced runs of one people row, 16 loads in flight per lane), convert to one 32-byte record (px, py, vx, vy)
// per (agent, step) at record index a * T + t in LDS (ag), and compute per step the bit mask of valid agents (vmask)
// and the agent-angle tag (aa). Executed by all W lanes of the slot.
template <int W>
__device__ inline void stage_people(KParamsK kp, int scene, int sl, double* ag, unsigned long long* vmask, double* aa) {
  const auto& k = *kp;
  const int T = k.T, N = k.N;
  const size_t s = scene;
  const double x0 = k.pose0[3 * s], y0 = k.pose0[3 * s + 1], yaw0 = k.pose0[3 * s + 2];
  const double* ppl = k.people + s * (size_t)(T + 1) * 6 * N;
  const int TN = T * N;
  // people_proj[t+1] field f agent a is at ((t+1)*6 + f)*N + a. Element (a, t) lands at a*T + t.
  for (int e0 = sl; e0 < TN; e0 += 4 * W) {
    double gx[4], gy[4], gyaw[4], glv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int e = min(e0 + u * W, TN - 1);
      const int t = e / N, a = e - t * N;
      const double* f = ppl + (size_t)(t + 1) * 6 * N + a;
      gx[u] = f[0]; gy[u] = f[N]; gyaw[u] = f[2 * N]; glv[u] = f[4 * N];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int e = e0 + u * W;
      if (e < TN) {
        const int t = e / N, a = e - t * N;
        double sn, cs;
        if (__builtin_expect(!(fabs(gyaw[u]) <= 1e5), 0)) sincos(gyaw[u], &sn, &cs);
        else sincos_tab(&k.mt, gyaw[u], &sn, &cs);
        const int q = a * T + t;
        v4d rec = {gx[u], gy[u], glv[u] * cs, glv[u] * sn};  // aVel, social_work:187-188
        reinterpret_cast<v4d*>(ag)[q] = rec;
      }
    }
  }
  if (sl < T) {
    const double* f = ppl + (size_t)(sl + 1) * 6 * N;
    unsigned long long m = 0;
    double aa_target = kNoTarget;
    // a7 AgentAngle tag: depends on constants only (critics/agent_angle_cost_function.hpp:130-190)
    int closest = -1;
    double best = INFINITY;
    for (int a0 = 0; a0 < N; a0 += 4) {  // loads of four agents in flight at a time
      double ft[4], fx[4], fy[4], fl[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int a = min(a0 + u, N - 1);
        ft[u] = f[3 * N + a]; fx[u] = f[a]; fy[u] = f[N + a]; fl[u] = f[4 * N + a];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int a = a0 + u;
        if (a < N) {
          if (ft[u] != -1.0) m |= (1ull << a);  // social_work:175
          const double ddx = fx[u] - x0, ddy = fy[u] - y0;
          const double d2 = ddx * ddx + ddy * ddy;
          if (d2 < best && fl[u] > 0.05) { best = d2; closest = a; }
        }
      }
    }
    if (closest >= 0 && !(best > 4.0)) {
      const double ax = f[closest], ay = f[N + closest], ayaw = f[2 * N + closest];
      const double agent_angle_initial = atan2(ay - y0, ax - x0);
      // atan2(sin u, cos u) of the reference (:148-152) restated as the range reduction wrap_angle(u)
      const double heading_diff = wrap_angle(ayaw - yaw0);
      const double rel = wrap_angle(agent_angle_initial - yaw0);
      const double kThr = M_PI / 6.0, kUp = 5 * M_PI / 6.0;
      if (heading_diff <= -kUp || heading_diff >= kThr) {
        if (!(rel < 0.0)) aa_target = yaw0 + (-(M_PI / 6.0));
      } else {
        if (!(rel > 0.0)) aa_target = yaw0 + (M_PI / 6.0);
      }
    }
    vmask[sl] = m;
    aa[sl] = aa_target;
  }
}

// The horizon of the slot's scene: rollout steps T, control horizon CH = min(control_horizon, T), block length
// bl = min(parameter_block_length, CH), index of the last parameter block, feasibility rows and bounded blocks
// (src/optimizer.cpp:248-249, 364, 373). kVT = false: one T per batch, everything is a launch constant (scalar
// registers) and the last block is NB - 1. kVT = true (smpc_scene_batch.T_scene): per scene, kept in LDS by load_scene();
// a scene with fewer blocks than the batch's NB keeps its surplus parameters at exactly zero — their Jacobian columns
// are zero, the damped system is block diagonal and every sum gains exact zeros, so the iterates of the real
// parameters are those of the smaller problem bit for bit.
struct Horizon { int T, CH, bl, blast, nfeas, nbounded; };

template <int NB, bool kVT> __device__ inline Horizon get_horizon(const Ctx& c) {
  Horizon h;
  if (!kVT) {
    const auto& k = *c.kp;
    h.T = k.T; h.CH = k.CH; h.bl = k.bl; h.blast = NB - 1; h.nfeas = k.nfeas; h.nbounded = k.nbounded;
  } else {
    const int* z = reinterpret_cast<const int*>(c.lds + c.L.hz);
    h.T = z[0]; h.CH = z[1]; h.bl = z[2]; h.blast = z[3]; h.nfeas = z[4]; h.nbounded = z[5];
  }
  return h;
}

// the block driving step sl: min(sl, CH - 1) / bl, as a count of block starts at or before it (no division)
template <int NB> __device__ inline int block_of_step(int sl, const Horizon& h) {
  const int t = min(sl, h.CH - 1);
  int b = 0;
#pragma unroll
  for (int q = 1; q < NB; ++q) b += (t >= q * h.bl) ? 1 : 0;
  return b;
}

// one past the last step block b drives: the next block's start, the horizon for the last block, nothing beyond it
__device__ inline int block_end(int b, const Horizon& h) {
  return b < h.blast ? (b + 1) * h.bl : (b == h.blast ? h.T : b * h.bl);
}

// Load the slot's scene constants and the per-step side data of its staged people block (valid masks, agent-angle
// tags) into LDS; the records themselves stay in global memory (c.ag). Executed by all W lanes of the slot (other
// slots may be masked off).
template <int W, bool kVT = false>
__device__ inline void load_scene(Ctx& c, int scene) {
  const auto& k = *c.kp;
  const int T = k.T, N = k.N, sl = c.sl;
  int Th = T;  // the scene's own horizon
  if (kVT) {
    const int Traw = k.T_scene ? k.T_scene[scene] : T;
    Th = min(max(Traw, 1), T);
    const int CH = min(k.prm.control_horizon, Th);
    const int bl = max(min(k.prm.parameter_block_length, CH), 1);
    int* z = reinterpret_cast<int*>(c.lds + c.L.hz);
    z[0] = Th; z[1] = CH; z[2] = bl; z[3] = (CH - 1) / bl;
    z[4] = max(min(CH / bl, Th) - 1, 0);  // src/optimizer.cpp:364
    z[5] = CH / bl;                       // :373
    z[6] = Traw < 1 ? 1 : 0;              // a path of fewer than two poses: Optimizer::optimize returns false (:158-162)
  }
  const size_t s = scene;
  c.scene = scene;
  c.has_people = (N > 0) && (k.has_people ? k.has_people[s] != 0 : true);
  c.ag = k.people_rec + s * (size_t)4 * T * (N > 0 ? N : 1);
  const double x0 = k.pose0[3 * s], y0 = k.pose0[3 * s + 1], yaw0 = k.pose0[3 * s + 2];
  const size_t cm = (size_t)k.size_x * k.size_y;
  c.map = k.costmap + (k.costmap_shared ? 0 : cm * s);
  const double* path_pts = k.path_pts + s * (T + 1) * 2;
  double* cst = c.lds + c.L.cst;
  cst[0] = x0; cst[1] = y0; cst[2] = yaw0;
  cst[3] = k.goal_yaw[s];
  cst[4] = k.costmap_origin[k.costmap_shared ? 0 : 2 * s];
  cst[5] = k.costmap_origin[k.costmap_shared ? 1 : 2 * s + 1];
  cst[6] = path_pts[2 * Th];  // final trajectorized point (src/optimizer.cpp:234-235)
  cst[7] = path_pts[2 * Th + 1];
  double* lanec = c.lds + c.L.lanec;
  if (sl < T) {
    double aa_target = kNoTarget;
    if (c.has_people) {
      const double* aux = k.people_aux + (s * T + sl) * 2;
      (c.lds + c.L.valid)[sl] = aux[0];  // the mask's bits travel in a double-sized slot
      aa_target = aux[1];
    }
    lanec[sl] = path_pts[2 * (sl + 1)];
    lanec[T + sl] = path_pts[2 * (sl + 1) + 1];
    lanec[2 * T + sl] = aa_target;
  }
}

// ------------------------------------------------------------------------------------------------
// The sweep (kernel K1's body): residuals + Jacobian rows + Gram at the slot's parameters xp[P] (LDS or global,
// slot-uniform). All 64 lanes of the wave call it together (each slot on its own scene). The Gram is returned
// reduced over the slot in every lane of the slot.
// When out_r / out_J are non-null (stand-alone K1) the rows are also written to HBM in the reference order.
// ------------------------------------------------------------------------------------------------
// kRows = false: the solve kernel's sweep, full Gram [J r]^T [J r] (structured, on the VALU), nothing written.
// kRows = true: the stand-alone K1 sweep; rows go to HBM, and of the Gram only its last column (J^T r and r^T r: the
// gradient and the cost smpc_eval_batch reports) is accumulated, on the VALU — c.wave_lds is then the row staging
// area of the critic-major store path (2 x T x P doubles per slot).
// need_rest (solve kernel only): called by every lane once the LAST column of the Gram — J^T r and r^T r: gradient and
// cost — is reduced and visible in LDS; the other columns (J^T J) are formed and reduced only if it returns true in some
// lane of the wave. A line-search sample that fails the Armijo test needs no more than that column (Ceres evaluates cost
// and gradient there, LineSearchFunction::Evaluate), and most sweeps of a solve are such samples. c.gram_full tells
// the caller what it got (wave-uniform). Every entry is summed in the same order whichever way it is produced.
struct NeedAll { __device__ inline bool operator()() const { return true; } };
template <int NB, int W, bool kRows, bool kVT = false, class NeedRest = NeedAll>
__device__ inline GramView sweep(Ctx& c, const double* xp, double* out_r, double* out_J, NeedRest need_rest = NeedRest()) {
  constexpr int P = 2 * NB;
  const auto& k = *c.kp;
  const int T = k.T, N = k.N, sl = c.sl;  // T: the batch's T = the stride of every per-step array
  const Horizon hz = get_horizon<NB, kVT>(c);
  const int Th = hz.T, CH = hz.CH, bl = hz.bl;  // Th: the scene's own rollout steps (== T unless kVT)
  const double dt = k.dt;
  double* cs_ = c.lds + c.L.cs;
  double* sn_ = cs_ + (T + 1);
  const double* cst = c.lds + c.L.cst;
  const int tl = min(sl, Th - 1);
  const int myb = block_of_step<NB>(sl, hz);  // block driving step sl
  const double vb = xp[2 * myb], wb = xp[2 * myb + 1];

  // staged people records of this lane's step: agent a at agr[a * T]; the first two are requested now, so that their
  // latency (HBM in the stand-alone K1 kernel) passes behind the rollout
  const v4d* agr = reinterpret_cast<const v4d*>(c.ag) + tl;
  v4d rec0 = {0.0, 0.0, 0.0, 0.0}, rec1 = {0.0, 0.0, 0.0, 0.0};
  if (c.has_people) { rec0 = agr[0]; rec1 = agr[(N > 1 ? 1 : 0) * T]; }

  // ---- a1 rollout (update_state.hpp:37-63), block-structured. The recurrences are sums, so they are evaluated as
  // sums: theta_t = yaw0 + sum_b (w_b dt) * (steps of block b before t), and the positions through inclusive prefix
  // sums over the lanes, restarted at every block boundary, of cos / sin(theta_j) and (j - block start) cos / sin:
  // every partial sum the pose and its sensitivities need is then one scan entry (the lane's own for its block, the
  // entry of a block's last step for the blocks before it) — no differences of long sums. (The reference adds the
  // terms one step at a time; the orders differ by rounding only, a few ulp.)
  double th = cst[2];
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    const int start = b * bl;
    const int len = block_end(b, hz) - start;
    const int cnt = min(max(sl - start, 0), len);  // steps j < sl that block b drives
    th = fma(xp[2 * b + 1] * dt, (double)cnt, th);
  }
  const double th1 = th + wb * dt;  // theta_{sl+1}
  double* scan_ = c.lds + c.L.inc;  // [4][T+1]: block-wise inclusive scans C, S, JC, JS over j = 0..T
  const int seg_start = myb * bl;
  {
    double sn, cs;
    // headings of a rollout are modest numbers; the library routine (Payne-Hanek reduction) only for a lane that holds a
    // heading beyond the two-part Cody-Waite range (an unbounded last parameter block can produce one)
    if (__builtin_expect(!(fabs(th) <= 1e5), 0)) sincos(th, &sn, &cs);
    else sincos_tab(&k.mt, th, &sn, &cs);
    const double fj = (double)(sl - seg_start);
    double sC = cs, sS = sn, sJC = fj * cs, sJS = fj * sn;
#pragma unroll
    for (int off = 1; off < W; off <<= 1) {
      const double uC = __shfl_up(sC, off, W), uS = __shfl_up(sS, off, W);
      const double uJC = __shfl_up(sJC, off, W), uJS = __shfl_up(sJS, off, W);
      const double m = (sl - off >= seg_start) ? 1.0 : 0.0;  // the source lane belongs to this lane's block
      sC = fma(m, uC, sC); sS = fma(m, uS, sS); sJC = fma(m, uJC, sJC); sJS = fma(m, uJS, sJS);
    }
    if (sl <= T) {
      cs_[sl] = cs; sn_[sl] = sn;
      scan_[sl] = sC; scan_[(T + 1) + sl] = sS; scan_[2 * (T + 1) + sl] = sJC; scan_[3 * (T + 1) + sl] = sJS;
    }
  }
  wave_lds_fence();
  SMPC_STAMP(c, 1);
  // x, y of pose_{sl+1} = pose0 + sum_b v_b dt * (sum of cos / sin(theta_j) over the steps j <= sl of block b)
  double X = cst[0], Y = cst[1];
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    const int end = block_end(b, hz);
    const int idx = (b < myb) ? end - 1 : tl;   // a finished block: its last step; the lane's own block: the lane
    const double m = (b <= myb) ? 1.0 : 0.0;
    const double vdt = xp[2 * b] * dt * m;
    X = fma(vdt, scan_[idx], X); Y = fma(vdt, scan_[(T + 1) + idx], Y);
  }
  const int t1 = min(sl + 1, Th);
  const double c1 = cs_[t1], s1 = sn_[t1];  // heading of the residual's pose
  // a5 obstacle: the costmap patch under the front point is requested now and used after the agent loop
  // cell coordinates of the front point: (front - origin) / resolution (critics/obstacle_cost_function.hpp:154-158) as a
  // product with the host-rounded reciprocal (within an ulp of the quotient; the interpolant is C1, so an ulp of its
  // argument is an ulp of its value)
  const bool wide_map = k.size_x >= 4;
  CostPatch patch;
  bool patch_interior = false;  // wave-uniform
  if (wide_map) {
    const double ob_ic = (X + 0.25 * c1 - cst[4]) * k.inv_resolution, ob_ir = (Y + 0.25 * s1 - cst[5]) * k.inv_resolution;
    patch_interior = __all(bicubic_interior(k.size_x, k.size_y, ob_ir, ob_ic));
    if (patch_interior) bicubic_fetch<true>(c.map, k.size_x, k.size_y, ob_ir, ob_ic, patch);
    else bicubic_fetch<false>(c.map, k.size_x, k.size_y, ob_ir, ob_ic, patch);
  }

  SMPC_STAMP(c, 2);
  // ---- a3 social work + a4 proxemics: walk the agents of step sl
  // what leaves this block is the finished social-work critic (residual and state-space gradient) and the nearest
  // valid agent of the proxemics critic: eight values across the divergent branch instead of the 23 running sums
  double sw_r = 0.0, sw_gx = 0.0, sw_gy = 0.0, sw_gt = 0.0, sw_gv = 0.0;
  double pbest = 1.7976931348623157e308, pdx = 0.0, pdy = 0.0;
  if (c.has_people) {
    double soc[kSoc];
#pragma unroll
    for (int i = 0; i < kSoc; ++i) soc[i] = 0.0;
    double su[4] = {0.0, 0.0, 0.0, 0.0};  // sums over valid agents of dF/du: (fx,ux) (fy,ux) (fx,uy) (fy,uy)
    double qh[4] = {0.0, 0.0, 0.0, 0.0};  // sums over pairs of F . dF/d(x, y, ux, uy), agent side
    const MathTabP mt = &k.mt;
    const double* atab = c.atab;
    const unsigned long long* vmask = reinterpret_cast<const unsigned long long*>(c.lds + c.L.valid);
    const unsigned long long vm = vmask[tl];
    const double rvx = vb * c1, rvy = vb * s1;  // meVel, social_work:170-171
    int pidx = 0;  // proxemics: index of the nearest valid agent (first minimum wins: std::min on duals)
    bool redo = false;  // some pair of this lane is not a regular one
    // One robot-agent pair: the robot of a step at (px, py) with velocity (pvx, pvy) against agent a of that step.
    // One evaluation serves both directions: the force on the robot from the agent (:125; valid agents only, :175)
    // is F(diff, u), diff = robot - agent, u = robotVel - agentVel; the force on the agent from the robot
    // (:137-143; every column, phantoms included) is F(-diff, -u) = -F with the same derivatives, so |.|^2 and its
    // gradient are those of F.
    auto pair = [&](const v4d& rec, int a, bool valid, double px, double py, double pvx, double pvy) {
      const double apx = rec[0], apy = rec[1], awx = rec[2], awy = rec[3];
      const double dx = px - apx, dy = py - apy;
      const double d2 = fma(dx, dx, dy * dy);
      const bool nearer = valid & (d2 < pbest);
      pbest = nearer ? d2 : pbest;
      pidx = nearer ? a : pidx;
      const Force F = pair_force(mt, atab, dx, dy, pvx - awx, pvy - awy);
      redo |= F.special | (d2 < 1e-12);  // |diff| < 1e-6 -> diff := (1e-6, 0), :181-184, breaks the symmetry above
      const double m = valid ? 1.0 : 0.0;
      soc[0] = fma(m, F.fx, soc[0]); soc[1] = fma(m, F.fy, soc[1]);
      soc[2] = fma(m, F.dfx_dx, soc[2]); soc[3] = fma(m, F.dfy_dx, soc[3]);
      soc[4] = fma(m, F.dfx_dy, soc[4]); soc[5] = fma(m, F.dfy_dy, soc[5]);
      // derivatives with respect to the robot's (theta, v) are one rotation of the u-derivatives that is the same
      // for every agent of this step (u = v (cos theta, sin theta) - agentVel): sum the u-derivatives, rotate once
      // after the loop
      su[0] = fma(m, F.dfx_dux, su[0]); su[1] = fma(m, F.dfy_dux, su[1]);
      su[2] = fma(m, F.dfx_duy, su[2]); su[3] = fma(m, F.dfy_duy, su[3]);
      soc[10] = fma(F.fx, F.fx, fma(F.fy, F.fy, soc[10]));
      qh[0] = fma(F.fx, F.dfx_dx, fma(F.fy, F.dfy_dx, qh[0]));
      qh[1] = fma(F.fx, F.dfx_dy, fma(F.fy, F.dfy_dy, qh[1]));
      qh[2] = fma(F.fx, F.dfx_dux, fma(F.fy, F.dfy_dux, qh[2]));
      qh[3] = fma(F.fx, F.dfx_duy, fma(F.fy, F.dfy_duy, qh[3]));
    };
    const int A = (W == 64) ? k.hp_A : N;  // agents the owner lane walks itself (helper_owner_agents(): N without helpers)
    if (W == 64 && A < N) {
      // ---- with helper lanes (see helper_owner_agents()): lanes sl >= T walk agents A .. N-1 of the steps h, h + R, ...
      const int NA = N - A, R = W - T;
      double* stp = c.lds + c.L.stepst;
      double* part = c.lds + c.L.part;
      const bool helper = sl >= T;
      if (sl < Th) { stp[sl] = X; stp[T + sl] = Y; stp[2 * T + sl] = rvx; stp[3 * T + sl] = rvy; }
      wave_lds_fence();
      // the step this lane currently works for, and what a pair evaluation needs of it
      int t_cur = helper ? sl - T : tl;
      bool unit_ok = helper ? (t_cur < Th) : (sl < Th);
      double cX = X, cY = Y, cvx = rvx, cvy = rvy;
      unsigned long long cvm = vm;
      if (helper) {
        const int ts = min(t_cur, Th - 1);
        cX = stp[ts]; cY = stp[T + ts]; cvx = stp[2 * T + ts]; cvy = stp[3 * T + ts]; cvm = vmask[ts];
      }
      int a_cur = helper ? A : 0, k_in = 0;
      const v4d* recs = reinterpret_cast<const v4d*>(c.ag);  // record of agent a at step t: recs[a * T + t]
      v4d rec_next = recs[a_cur * T + min(t_cur, Th - 1)];
      for (int it = 0; it < A; ++it) {
        const v4d rec = rec_next;
        // where this lane goes next (a helper that has finished a unit moves R steps on), fetched one pair ahead
        const bool unit_end = helper && (k_in + 1 == NA);
        const int t_next = unit_end ? t_cur + R : t_cur;
        const int a_next = unit_end ? A : a_cur + 1;
        rec_next = recs[min(a_next, N - 1) * T + min(t_next, Th - 1)];
        if (unit_ok) pair(rec, a_cur, (cvm >> a_cur) & 1ull, cX, cY, cvx, cvy);
        if (unit_end) {
          if (unit_ok) {  // hand the unit's sums to the owner lane of its step
            double* pr = part + t_cur * kPart;
#pragma unroll
            for (int i = 0; i < 6; ++i) pr[i] = soc[i];
#pragma unroll
            for (int i = 0; i < 4; ++i) { pr[6 + i] = su[i]; pr[11 + i] = qh[i]; }
            pr[10] = soc[10];
            pr[15] = pbest;
            pr[16] = (double)(pidx + (redo ? 4096 : 0));
          }
#pragma unroll
          for (int i = 0; i < 6; ++i) soc[i] = 0.0;
#pragma unroll
          for (int i = 0; i < 4; ++i) { su[i] = 0.0; qh[i] = 0.0; }
          soc[10] = 0.0; pbest = 1.7976931348623157e308; pidx = 0; redo = false;
          unit_ok = t_next < Th;
          const int ts = min(t_next, Th - 1);
          cX = stp[ts]; cY = stp[T + ts]; cvx = stp[2 * T + ts]; cvy = stp[3 * T + ts]; cvm = vmask[ts];
          k_in = 0;
        } else {
          k_in += helper ? 1 : 0;
        }
        t_cur = t_next; a_cur = a_next;
      }
      wave_lds_fence();
      if (!helper && sl < Th) {  // the owner adds what the helper found for its step
        const double* pr = part + sl * kPart;
#pragma unroll
        for (int i = 0; i < 6; ++i) soc[i] += pr[i];
#pragma unroll
        for (int i = 0; i < 4; ++i) { su[i] += pr[6 + i]; qh[i] += pr[11 + i]; }
        soc[10] += pr[10];
        const int tag = (int)pr[16];
        // the helper's agents come after the owner's: a tie stays with the owner (std::min keeps the first minimum)
        if (pr[15] < pbest) { pbest = pr[15]; pidx = tag & 4095; }
        redo |= tag >= 4096;
      }
    } else {
      for (int a = 0; a < N; ++a) {
        // records are fetched two agents ahead (the first two before the rollout, above): HBM latency in K1, L2 in a solve
        const v4d rec = rec0;
        rec0 = rec1;
        if (a + 2 < N) rec1 = agr[(a + 2) * T];
        pair(rec, a, (vm >> a) & 1ull, X, Y, rvx, rvy);
      }
    }
    {  // the constant factors pair_force() leaves out, once per step instead of once per pair
      const double kk = kPairForceK, kl = kPairForceK * kPairForceLambda, k2 = kPairForceK * kPairForceK,
                   k2l = kPairForceK * kPairForceK * kPairForceLambda;
#pragma unroll
      for (int i = 0; i < 6; ++i) soc[i] *= kk;
#pragma unroll
      for (int i = 0; i < 4; ++i) su[i] *= kl;
      soc[10] *= k2; qh[0] *= k2; qh[1] *= k2; qh[2] *= k2l; qh[3] *= k2l;
    }
    if (__builtin_expect(redo, 0)) {
      // Rare: this lane met a pair the fast form does not cover. It walks its agents again in the general form (both
      // directions evaluated where the symmetry does not hold), from cleared sums. Decided per lane: the result of a
      // scene must not depend on which other scene shares its wave.
#pragma unroll
      for (int i = 0; i < kSoc; ++i) soc[i] = 0.0;
#pragma unroll
      for (int i = 0; i < 4; ++i) { su[i] = 0.0; qh[i] = 0.0; }
      for (int a = 0; a < N; ++a) {
        const v4d rec = agr[a * T];
        const double apx = rec[0], apy = rec[1], awx = rec[2], awy = rec[3];
        const bool valid = (vm >> a) & 1ull;
        const double dx = X - apx, dy = Y - apy;
        const double d2 = dx * dx + dy * dy;
        const bool degenerate = d2 < 1e-12;
        if (valid) {
          const Force F = social_force_general(dx, dy, rvx - awx, rvy - awy);
          soc[0] += F.fx; soc[1] += F.fy;
          soc[2] += F.dfx_dx; soc[3] += F.dfy_dx; soc[4] += F.dfx_dy; soc[5] += F.dfy_dy;
          su[0] += F.dfx_dux; su[1] += F.dfy_dux; su[2] += F.dfx_duy; su[3] += F.dfy_duy;
          if (!degenerate) {
            soc[10] = fma(F.fx, F.fx, fma(F.fy, F.fy, soc[10]));
            qh[0] = fma(F.fx, F.dfx_dx, fma(F.fy, F.dfy_dx, qh[0]));
            qh[1] = fma(F.fx, F.dfx_dy, fma(F.fy, F.dfy_dy, qh[1]));
            qh[2] = fma(F.fx, F.dfx_dux, fma(F.fy, F.dfy_dux, qh[2]));
            qh[3] = fma(F.fx, F.dfx_duy, fma(F.fy, F.dfy_duy, qh[3]));
          }
        }
        if (!valid || degenerate) {
          const Force G = social_force_general(-dx, -dy, awx - rvx, awy - rvy);
          soc[10] += G.fx * G.fx + G.fy * G.fy;
          // d/d(robot x, y) = -d/d(diff), d/d(robot u) = -d/d(u): entered with the sign, rotated with the rest below
          qh[0] -= G.fx * G.dfx_dx + G.fy * G.dfy_dx;
          qh[1] -= G.fx * G.dfx_dy + G.fy * G.dfy_dy;
          qh[2] -= G.fx * G.dfx_dux + G.fy * G.dfy_dux;
          qh[3] -= G.fx * G.dfx_duy + G.fy * G.dfy_duy;
        }
      }
    }
    {
      const v4d prec = agr[pidx * T];
      pdx = X - prec[0]; pdy = Y - prec[1];
    }
    soc[6] = vb * (-s1 * su[0] + c1 * su[2]); soc[7] = vb * (-s1 * su[1] + c1 * su[3]);  // d(sum F)/d theta
    soc[8] = c1 * su[0] + s1 * su[2]; soc[9] = c1 * su[1] + s1 * su[3];                  // d(sum F)/d v
    soc[11] += 2.0 * qh[0];
    soc[12] += 2.0 * qh[1];
    soc[13] += 2.0 * (vb * (-s1 * qh[2] + c1 * qh[3]));
    soc[14] += 2.0 * (c1 * qh[2] + s1 * qh[3]);
    // a3 social work: w (|sum F|^2 + sum |G|^2 + 1e-6), critics/social_work_cost_function.hpp:125-147
    const double wsoc = k.prm.socialwork_w;
    const double wr = soc[0] * soc[0] + soc[1] * soc[1];
    sw_r = wsoc * (wr + soc[10] + 1e-6);
    sw_gx = wsoc * (2.0 * (soc[0] * soc[2] + soc[1] * soc[3]) + soc[11]);
    sw_gy = wsoc * (2.0 * (soc[0] * soc[4] + soc[1] * soc[5]) + soc[12]);
    sw_gt = wsoc * (2.0 * (soc[0] * soc[6] + soc[1] * soc[7]) + soc[13]);
    sw_gv = wsoc * (2.0 * (soc[0] * soc[8] + soc[1] * soc[9]) + soc[14]);
  }

  SMPC_STAMP(c, 3);
  // ---- sensitivities S of pose_{sl+1}, from the scans. K1 needs them for every row it forms, so right after the agent
  // loop; the solve kernel only for the final M^T A M, so after the critics (they stay out of the critics' register
  // budget, as the critics' sums stay out of theirs).
  double Sxv[NB], Syv[NB], Sxw[NB], Syw[NB], Sthw[NB];
  auto compute_sensitivities = [&]() {
  #pragma unroll
    for (int b = 0; b < NB; ++b) {
      const int start = b * bl;
      const int end = block_end(b, hz);
      const int idx = (b < myb) ? end - 1 : tl;
      const double m = (b <= myb) ? 1.0 : 0.0;
      const double aC = m * scan_[idx], aS = m * scan_[(T + 1) + idx];
      const double aJC = m * scan_[2 * (T + 1) + idx], aJS = m * scan_[3 * (T + 1) + idx];  // sums of (j - start) cos / sin
      const double vdt = xp[2 * b] * dt;
      Sxv[b] = dt * aC;
      Syv[b] = dt * aS;
      // d theta_j / d w_b = dt (j - start) inside block b; = dt * bl for every later step
      Sxw[b] = -vdt * dt * aJS;
      Syw[b] = vdt * dt * aJC;
  #pragma unroll
      for (int q = 0; q < b; ++q) {
        Sxw[q] = fma(-vdt * dt * (double)bl, aS, Sxw[q]);
        Syw[q] = fma(vdt * dt * (double)bl, aC, Syw[q]);
      }
      const int cnt = min(max(sl + 1 - start, 0), end - start);
      Sthw[b] = dt * (double)max(cnt, 0);
    }
  };
  if (kRows) compute_sensitivities();
  // K1 with seven or more parameter blocks: the 5 NB sensitivities of a lane (100 registers at NB = 10) are parked in LDS
  // behind the row staging blocks, lane index fastest, and read back entry by entry while the rows are formed — with
  // them in registers the kernel needed 256 VGPRs plus up to 110 AGPR copies
  constexpr bool kSensInLds = kRows && NB > kSensInRegsMaxBlocks;
  double* sens_lds = c.wave_lds + (kWave / W) * (2 * T * P) + c.slot * (5 * NB * W) + sl;
  if (kSensInLds) {
#pragma unroll
    for (int q = 0; q < NB; ++q) {
      sens_lds[(0 * NB + q) * W] = Sxv[q]; sens_lds[(1 * NB + q) * W] = Syv[q]; sens_lds[(2 * NB + q) * W] = Sxw[q];
      sens_lds[(3 * NB + q) * W] = Syw[q]; sens_lds[(4 * NB + q) * W] = Sthw[q];
    }
    wave_lds_fence();
  }

  SMPC_STAMP(c, 4);
  // ---- per-step critics: residual r and its state-space gradient (gx, gy, gth; gv = direct derivative with respect
  // to the linear velocity of block myb). Row of the Jacobian = gradient x M, M = [S; selector] (4 x P), the same M
  // for every critic of the step.
  //   kRows (K1): the rows are formed and written to HBM; of the Gram only the last column is kept (VALU).
  //   solve:      rows are never formed. Per lane A = sum_k g_k^T g_k (4 x 4), b = sum_k g_k r_k, cc = sum_k r_k^2 are
  //               accumulated over the step's critics (most gradients have one to three non-zero components), then
  //               the lane's share of [J r]^T [J r] is M^T A M, M^T b, cc — 1/3 of the multiply-adds of forming and
  //               contracting eight P-wide rows, and no FP64 MFMA (on this part it runs at the FP64 VALU rate on the
  //               same pipe: measured slower than plain VALU accumulation).
  constexpr int Q = P + 1;
  const auto& w = k.prm;
  const bool lane_live = sl < Th;
  const bool people = c.has_people;
  const int rows_per_step = people ? 8 : 5;
  const int row_base = rows_per_step * sl + min(max(sl - 1, 0), hz.nfeas);
  double gcol[Q];      // kRows: last column of the Gram only
  double Axx = 0.0, Axy = 0.0, Axt = 0.0, Axv = 0.0, Ayy = 0.0, Ayt = 0.0, Ayv = 0.0, Att = 0.0, Atv = 0.0, Avv = 0.0;
  double bx = 0.0, by = 0.0, bt = 0.0, bv = 0.0, cc = 0.0;
  double* stage = c.wave_lds + c.slot * (2 * T * P);  // kRows: two row blocks of this slot, used alternately
  const bool critic_major = kRows && k.e_row_order == 1;
  if (kRows) {
#pragma unroll
    for (int q = 0; q < Q; ++q) gcol[q] = 0.0;
  }
  // One row of the Jacobian = gradient x M, formed two entries (one parameter block) at a time: each pair goes to its
  // destination and into the last Gram column at once, so no whole row is ever held in registers (with ten parameter
  // blocks a row is 40 registers on top of the 100 of the sensitivities).
  auto emit = [&](int local, bool live, bool slot_on, double r, double gx, double gy, double gth, double gv) {  // kRows only
    const double rl = live ? r : 0.0;
    double* blk = stage + (local & 1) * (T * P);
    const int rowi = row_base + local;
    const bool to_lds = critic_major && sl < T && slot_on;   // every row of the block, the all-zero rows of steps beyond
                                                             // the scene's horizon included
    const bool to_mem = !critic_major && live && out_J;
#pragma unroll
    for (int q = 0; q < NB; ++q) {
      const double sxv = kSensInLds ? sens_lds[(0 * NB + q) * W] : Sxv[q], syv = kSensInLds ? sens_lds[(1 * NB + q) * W] : Syv[q];
      const double sxw = kSensInLds ? sens_lds[(2 * NB + q) * W] : Sxw[q], syw = kSensInLds ? sens_lds[(3 * NB + q) * W] : Syw[q];
      const double sthw = kSensInLds ? sens_lds[(4 * NB + q) * W] : Sthw[q];
      double e0 = gx * sxv + gy * syv + ((q == myb) ? gv : 0.0);
      double e1 = gx * sxw + gy * syw + gth * sthw;
      e0 = live ? e0 : 0.0; e1 = live ? e1 : 0.0;
      gcol[2 * q] = fma(e0, rl, gcol[2 * q]);
      gcol[2 * q + 1] = fma(e1, rl, gcol[2 * q + 1]);
      v2d pr = {e0, e1};
      // critic-major: row (critic `local`, step sl) lives at index local * T + sl, the T rows of one critic are one
      // contiguous block of T * P doubles. They pass through LDS so that every store instruction writes whole runs of
      // consecutive 16-byte pieces (lane-strided 48-byte rows touch 64 different lines per instruction and leave
      // partially written lines behind: measured 1.49 x write amplification in round 1).
      if (to_lds) reinterpret_cast<v2d*>(blk + sl * P)[q] = pr;
      if (to_mem) reinterpret_cast<v2d*>(out_J + (size_t)rowi * P)[q] = pr;
    }
    gcol[P] = fma(rl, rl, gcol[P]);
    if (critic_major) {
      wave_lds_fence();
      if (out_J && slot_on) {
        v2d* dst = reinterpret_cast<v2d*>(out_J + (size_t)local * T * P);
        const v2d* src = reinterpret_cast<const v2d*>(blk);
        for (int i = sl; i < T * NB; i += W) dst[i] = src[i];
      }
      if (out_r && sl < T && slot_on) out_r[local * T + sl] = rl;
      // no second fence: the next critic writes the other block, and the one after that comes behind this
      // critic's reads in program order with a fence in between
    } else if (live && out_r) {
      out_r[rowi] = r;
    }
  };
  if (__any(people)) {
    const bool live = lane_live && people;
    // a7 agent angle
    {
      double r = 0.0, gth = 0.0;
      const double aa_target = (c.lds + c.L.lanec)[2 * T + tl];
      if (aa_target != kNoTarget) {
        const double ad = wrap_angle(th1 - aa_target);
        r = w.agent_angle_w * (ad * ad);
        gth = w.agent_angle_w * 2.0 * ad;
      }
      if (kRows) emit(0, live, people, r, 0.0, 0.0, gth, 0.0);
      else {  // (a slot without people beside one with people has no steering target: r = gth = 0 already)
        Att = fma(gth, gth, Att); bt = fma(gth, r, bt); cc = fma(r, r, cc);
      }
    }
    // a3 social work (finished inside the agent block above; zero for a slot without people)
    {
      const double r = sw_r, gx = sw_gx, gy = sw_gy, gt = sw_gt, gv = sw_gv;
      if (kRows) emit(1, live, people, r, gx, gy, gt, gv);
      else {
        Axx = fma(gx, gx, Axx); Axy = fma(gx, gy, Axy); Axt = fma(gx, gt, Axt); Axv = fma(gx, gv, Axv);
        Ayy = fma(gy, gy, Ayy); Ayt = fma(gy, gt, Ayt); Ayv = fma(gy, gv, Ayv);
        Att = fma(gt, gt, Att); Atv = fma(gt, gv, Atv); Avv = fma(gv, gv, Avv);
        bx = fma(gx, r, bx); by = fma(gy, r, by); bt = fma(gt, r, bt); bv = fma(gv, r, bv); cc = fma(r, r, cc);
      }
    }
    // a4 proxemics: w alpha exp(-min_a d^2 / d0^2) over valid agents
    {
      const double e = 3.0 * exp_tab(&k.mt, -pbest / (0.5 * 0.5));
      double r = w.proxemics_w * e;
      double gx = r * (-2.0 * pdx / (0.5 * 0.5)), gy = r * (-2.0 * pdy / (0.5 * 0.5));
      if (people && pbest == 1.7976931348623157e308) {
        // no valid agent: the reference's dual evaluation gives (-max / d0^2) = -inf and inf * 0 = NaN tangents
        // (critics/proxemics_cost_function.hpp:127,147) -> Ceres rejects the evaluation. Mirror it.
        gx = gy = __longlong_as_double(0x7ff8000000000000ll);
      }
      if (kRows) emit(2, live, people, r, gx, gy, 0.0, 0.0);
      else {  // (a slot without people: pbest is still the largest double, exp gives r = 0 and with it gx = gy = 0)
        Axx = fma(gx, gx, Axx); Axy = fma(gx, gy, Axy); Ayy = fma(gy, gy, Ayy);
        bx = fma(gx, r, bx); by = fma(gy, r, by); cc = fma(r, r, cc);
      }
    }
  }
  SMPC_STAMP2(c, 0);  // people critics (agent angle, social combine, proxemics)
  const int o5 = people ? 3 : 0;
  // a6 velocity
  {
    double r = 0.0, gv = 0.0;
    if (sl < CH) { const double d = w.desired_linear_vel - vb; r = w.velocity_w * d * d; gv = -2.0 * w.velocity_w * d; }
    if (kRows) emit(o5 + 0, lane_live, true, r, 0.0, 0.0, 0.0, gv);
    else { Avv = fma(gv, gv, Avv); bv = fma(gv, r, bv); cc = fma(r, r, cc); }
  }
  // a8 goal align
  {
    const double a = wrap_angle(cst[3] - th1);
    const double r = w.goal_align_w * a * a, gth = -2.0 * w.goal_align_w * a;
    if (kRows) emit(o5 + 1, lane_live, true, r, 0.0, 0.0, gth, 0.0);
    else { Att = fma(gth, gth, Att); bt = fma(gth, r, bt); cc = fma(r, r, cc); }
  }
  // a2 distance (path follow -> final point; path align -> point sl+1)
  {
    const double ddx = X - cst[6], ddy = Y - cst[7], q2 = ddx * ddx + ddy * ddy;
    const double r = w.distance_w * q2 * q2, gx = 4.0 * w.distance_w * q2 * ddx, gy = 4.0 * w.distance_w * q2 * ddy;
    if (kRows) emit(o5 + 2, lane_live, true, r, gx, gy, 0.0, 0.0);
    else {
      Axx = fma(gx, gx, Axx); Axy = fma(gx, gy, Axy); Ayy = fma(gy, gy, Ayy);
      bx = fma(gx, r, bx); by = fma(gy, r, by); cc = fma(r, r, cc);
    }
  }
  {
    const double* lanec = c.lds + c.L.lanec;
    const double ddx = X - lanec[tl], ddy = Y - lanec[T + tl], q2 = ddx * ddx + ddy * ddy;
    const double r = w.angle_w * q2 * q2, gx = 4.0 * w.angle_w * q2 * ddx, gy = 4.0 * w.angle_w * q2 * ddy;
    if (kRows) emit(o5 + 3, lane_live, true, r, gx, gy, 0.0, 0.0);
    else {
      Axx = fma(gx, gx, Axx); Axy = fma(gx, gy, Axy); Ayy = fma(gy, gy, Ayy);
      bx = fma(gx, r, bx); by = fma(gy, r, by); cc = fma(r, r, cc);
    }
  }
  SMPC_STAMP2(c, 1);  // velocity, goal, 2 x distance
  // a5 obstacle
  {
    const double inv_res = k.inv_resolution;
    // the same expressions as at the fetch (recomputed rather than kept in registers across the agent loop)
    const double ob_ic = (X + 0.25 * c1 - cst[4]) * inv_res, ob_ir = (Y + 0.25 * s1 - cst[5]) * inv_res;
    double f, dfdr, dfdc;
    if (patch_interior) bicubic_eval<true>(patch, ob_ir, ob_ic, f, dfdr, dfdc);
    else if (wide_map) bicubic_eval<false>(patch, ob_ir, ob_ic, f, dfdr, dfdc);
    else bicubic(c.map, k.size_x, k.size_y, ob_ir, ob_ic, f, dfdr, dfdc);  // maps narrower than one patch: byte by byte
    const double r = w.obstacle_w * f;
    const double gx = w.obstacle_w * dfdc * inv_res, gy = w.obstacle_w * dfdr * inv_res;
    const double gth = w.obstacle_w * (dfdc * (-0.25 * s1) + dfdr * (0.25 * c1)) * inv_res;
    if (kRows) emit(o5 + 4, lane_live, true, r, gx, gy, gth, 0.0);
    else {
      Axx = fma(gx, gx, Axx); Axy = fma(gx, gy, Axy); Axt = fma(gx, gth, Axt);
      Ayy = fma(gy, gy, Ayy); Ayt = fma(gy, gth, Ayt); Att = fma(gth, gth, Att);
      bx = fma(gx, r, bx); by = fma(gy, r, by); bt = fma(gth, r, bt); cc = fma(r, r, cc);
    }
  }
  SMPC_STAMP2(c, 2);  // obstacle (bicubic gather)
  GramView view;
  double* gt = c.lds + c.L.gram;
  view.base = gt;
  view.ld = Q;
  if (kRows) {
    // a9 velocity feasibility between blocks sl and sl-1 (src/optimizer.cpp:364-370); the row follows step sl
    if (k.nfeas > 0) {
      const bool live = lane_live && sl >= 1 && sl <= hz.nfeas;
      double row[P];
#pragma unroll
      for (int q = 0; q < P; ++q) row[q] = 0.0;
      double r = 0.0;
#pragma unroll
      for (int q = 1; q < NB; ++q) {
        if (q == sl) {
          const double lin = xp[2 * q] - xp[2 * q - 2], ang = xp[2 * q + 1] - xp[2 * q - 1];
          r = w.velocity_feasibility_w * lin * lin + w.velocity_feasibility_w * ang * ang;
          row[2 * q] = 2.0 * w.velocity_feasibility_w * lin;
          row[2 * q - 2] = -2.0 * w.velocity_feasibility_w * lin;
          row[2 * q + 1] = 2.0 * w.velocity_feasibility_w * ang;
          row[2 * q - 1] = -2.0 * w.velocity_feasibility_w * ang;
        }
      }
      if (live) {
        const int rowi = critic_major ? rows_per_step * T + (sl - 1) : row_base + rows_per_step;
        if (out_r) out_r[rowi] = r;
        if (out_J) {
#pragma unroll
          for (int q = 0; q < P; ++q) out_J[(size_t)rowi * P + q] = row[q];
        }
      }
      {
        const double rl = live ? r : 0.0;
#pragma unroll
        for (int q = 0; q < P; ++q) gcol[q] = fma(live ? row[q] : 0.0, rl, gcol[q]);
        gcol[P] = fma(rl, rl, gcol[P]);
      }
    }
#pragma unroll
    for (int a = 0; a < Q; ++a) {
      const double v = slot_sum<W>(gcol[a]);
      gt[a * Q + P] = v;
      gt[P * Q + a] = v;
    }
  } else {
    compute_sensitivities();
    // ---- the lane's share of the Gram: H = M^T A M, g = M^T b, cc; summed over the slot's lanes through LDS.
    if (!lane_live) {  // lanes beyond the horizon carry nothing (their A may hold anything, NaN included)
      Axx = Axy = Axt = Axv = Ayy = Ayt = Ayv = Att = Atv = Avv = 0.0;
      bx = by = bt = bv = cc = 0.0;
    }
    // a9 velocity feasibility rows (src/optimizer.cpp:364-370): row q (between blocks q and q-1, 1 <= q <= nfeas) has
    // four non-zero entries; the rows go to LDS as they are and their outer products are added after the lane sum.
    double* frow = c.wave_lds + c.slot * ((NB > 1 ? NB - 1 : 1) * Q);
    if (sl >= 1 && sl <= hz.nfeas) {
      const double lin = xp[2 * sl] - xp[2 * sl - 2], ang = xp[2 * sl + 1] - xp[2 * sl - 1];
      const double wf = w.velocity_feasibility_w;
      double* fr = frow + (sl - 1) * Q;
#pragma unroll
      for (int q = 0; q < Q; ++q) fr[q] = 0.0;
      fr[2 * sl - 2] = -2.0 * wf * lin; fr[2 * sl - 1] = -2.0 * wf * ang;
      fr[2 * sl] = 2.0 * wf * lin; fr[2 * sl + 1] = 2.0 * wf * ang;
      fr[P] = wf * lin * lin + wf * ang * ang;
    }
    double* red_base = c.lds + c.L.cs;  // over the cos / sin block and the scans (dead by now) and the tail behind them
    double* red = red_base + sl * (kGramChunk + 1);
    double hv[kGramChunk];
    int cnt = 0, chunk_base = 0;  // compile-time after unrolling
    auto flush = [&](int n, int base, int col_lo, int col_hi) {
      wave_lds_fence();  // the previous chunk's sums have been read
#pragma unroll
      for (int i = 0; i < kGramChunk; ++i) if (i < n) red[i] = hv[i];
      wave_lds_fence();
      // two levels: lane (part, v) = (sl / 16, sl % 16) adds up value v of the 16 lanes of its part, then the W / 16
      // parts are combined by butterfly shuffles — W / 16 times fewer additions per lane than one lane per value
      const int vsel = sl & (kGramChunk - 1), part = sl / kGramChunk;
      double tot = 0.0;
      {
        const double* col = red_base + (part * kGramChunk) * (kGramChunk + 1) + vsel;
#pragma unroll
        for (int l = 0; l < kGramChunk; ++l) tot += col[l * (kGramChunk + 1)];
      }
#pragma unroll
      for (int off = kGramChunk; off < W; off <<= 1) tot += __shfl_xor(tot, off, W);
      if (sl < n) {
        // packed (column-major upper triangle) index -> (a, b)
        const int pidx = base + sl;
        int bcol = col_lo;
#pragma unroll
        for (int cb = 1; cb < Q; ++cb) if (cb > col_lo && cb <= col_hi) bcol = (pidx >= cb * (cb + 1) / 2) ? cb : bcol;
        const int arow = pidx - bcol * (bcol + 1) / 2;
        for (int q = 0; q < hz.nfeas; ++q) tot = fma(frow[q * Q + arow], frow[q * Q + bcol], tot);
        gt[arow * Q + bcol] = tot;
        gt[bcol * Q + arow] = tot;
      }
    };
    int col_lo = 0;
    auto column = [&](int bq) {  // column bq of [J r]: 2q = v_q, 2q+1 = w_q, P = r; entries 0 .. bq (upper triangle)
      double Wx, Wy, Wt, Wv;  // column bq of A M (for bq = P: b itself)
      if (bq == P) { Wx = bx; Wy = by; Wt = bt; Wv = bv; }
      else if ((bq & 1) == 0) {
        const int q = bq >> 1;
        const double mq = (q == myb) ? 1.0 : 0.0;
        Wx = fma(Axx, Sxv[q], fma(Axy, Syv[q], Axv * mq));
        Wy = fma(Axy, Sxv[q], fma(Ayy, Syv[q], Ayv * mq));
        Wt = fma(Axt, Sxv[q], fma(Ayt, Syv[q], Atv * mq));
        Wv = fma(Axv, Sxv[q], fma(Ayv, Syv[q], Avv * mq));
      } else {
        const int q = bq >> 1;
        Wx = fma(Axx, Sxw[q], fma(Axy, Syw[q], Axt * Sthw[q]));
        Wy = fma(Axy, Sxw[q], fma(Ayy, Syw[q], Ayt * Sthw[q]));
        Wt = fma(Axt, Sxw[q], fma(Ayt, Syw[q], Att * Sthw[q]));
        Wv = fma(Axv, Sxw[q], fma(Ayv, Syw[q], Atv * Sthw[q]));
      }
#pragma unroll
      for (int aq = 0; aq <= bq; ++aq) {
        double h;
        if (aq == P) h = cc;
        else if ((aq & 1) == 0) {
          const int q = aq >> 1;
          const double mq = (q == myb) ? 1.0 : 0.0;
          h = fma(Sxv[q], Wx, fma(Syv[q], Wy, mq * Wv));
        } else {
          const int q = aq >> 1;
          h = fma(Sxw[q], Wx, fma(Syw[q], Wy, Sthw[q] * Wt));
        }
        if (cnt == 0) col_lo = bq;
        hv[cnt++] = h;
        if (cnt == kGramChunk) { flush(cnt, chunk_base, col_lo, bq); chunk_base += cnt; cnt = 0; }  // a full chunk
      }
    };
    // the last column first: gradient and cost, enough for a line-search sample that fails the Armijo test
    chunk_base = P * (P + 1) / 2;
    column(P);
    if (cnt > 0) { flush(cnt, chunk_base, col_lo, P); cnt = 0; }
    wave_lds_fence();
    c.gram_full = __any(need_rest());
    if (c.gram_full) {
      chunk_base = 0;
#pragma unroll
      for (int bq = 0; bq < P; ++bq) column(bq);
      if (cnt > 0) flush(cnt, chunk_base, col_lo, P - 1);
    }
  }
  wave_lds_fence();  // Gram visible to every lane of the slot; the cos/sin block may be rewritten by the next sweep
  SMPC_STAMP(c, 5);
  return view;
}

// A sweep result is usable iff every residual and Jacobian entry was finite: a non-finite entry makes the
// corresponding diagonal entry of the Gram (a sum of squares) non-finite.
template <int P> __device__ inline bool gram_finite(const GramView& g) {
  bool ok = true;
#pragma unroll
  for (int q = 0; q <= P; ++q) ok = ok && isfinite(g(q, q));
  return ok;
}

}  // namespace smpc
