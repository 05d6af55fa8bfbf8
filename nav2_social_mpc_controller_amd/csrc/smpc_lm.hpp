// smpc_lm.hpp — per-slot Levenberg-Marquardt state machine (the ceres::Solve call of reference
// src/optimizer.cpp:381, options :117-131) and the post-solve unpack (src/optimizer.cpp:390-446).
// Algorithm = Ceres' trust-region minimizer with bounds as specified in SURVEY.md Appendix A (A.4 .. A.11).
//
// A wave is a persistent "sweep engine": every trip of the main loop runs ONE sweep() for all slots of the wave
// at each slot's current trial point, then each slot advances its own LM state (phases below) and produces its
// next trial point, or finishes its scene and pulls the next one from the global scene queue. Slots never wait for
// each other: iteration counts and line-search lengths differ per scene, the sweep is the only shared code.
// Every LM quantity is uniform across the W lanes of a slot (computed redundantly, LM vectors parked in LDS).
#pragma once

#include "smpc_device.hpp"

namespace smpc {

#ifdef SMPC_STAMPS
// diagnostic build only: [0] bracketed_root<4> calls, [1] their iterations, [2] bracketed_root<3> calls, [3] iterations,
// [4] interpolations (cubic), [5] (quintic), [6] quintic shortcut taken, [7] generic fallback
__device__ unsigned long long g_ls_dbg[8];
#define SMPC_LS_COUNT(i, n) do { if ((threadIdx.x & 31) == 0) atomicAdd(&g_ls_dbg[i], (unsigned long long)(n)); } while (0)
#else
#define SMPC_LS_COUNT(i, n) do { } while (0)
#endif

// ---- small uniform helpers working on an LDS scratch area (dynamic indexing without private scratch) ----

// Solve A z = b with full pivoting, n <= 6, A row-major n x n in LDS (destroyed). Result in z (LDS).
__device__ inline void fullpiv_solve(double* A, double* b, int* perm, double* z, double* out, int n) {
  for (int i = 0; i < n; ++i) perm[i] = i;
  for (int kk = 0; kk < n; ++kk) {
    int pr = kk, pc = kk;
    double best = -1.0;
    for (int i = kk; i < n; ++i)
      for (int j = kk; j < n; ++j) {
        const double v = fabs(A[i * n + j]);
        if (v > best) { best = v; pr = i; pc = j; }
      }
    if (best == 0.0) { for (int i = kk; i < n; ++i) b[i] = 0.0; break; }
    if (pr != kk) {
      for (int j = 0; j < n; ++j) { const double t = A[pr * n + j]; A[pr * n + j] = A[kk * n + j]; A[kk * n + j] = t; }
      const double t = b[pr]; b[pr] = b[kk]; b[kk] = t;
    }
    if (pc != kk) {
      for (int i = 0; i < n; ++i) { const double t = A[i * n + pc]; A[i * n + pc] = A[i * n + kk]; A[i * n + kk] = t; }
      const int t = perm[pc]; perm[pc] = perm[kk]; perm[kk] = t;
    }
    for (int i = kk + 1; i < n; ++i) {
      const double f = A[i * n + kk] / A[kk * n + kk];
      for (int j = kk; j < n; ++j) A[i * n + j] -= f * A[kk * n + j];
      b[i] -= f * b[kk];
    }
  }
  for (int i = n - 1; i >= 0; --i) {
    if (A[i * n + i] == 0.0) { z[i] = 0.0; continue; }
    double v = b[i];
    for (int kk = i + 1; kk < n; ++kk) v -= A[i * n + kk] * z[kk];
    z[i] = v / A[i * n + i];
  }
  for (int i = 0; i < n; ++i) out[perm[i]] = z[i];
}

__device__ inline double eval_poly(const double* p, int ncoef, double x) {
  double v = 0.0;
  for (int i = 0; i < ncoef; ++i) v = v * x + p[i];
  return v;
}

// Real parts of all roots of the polynomial p (highest degree first, ncoef coefficients) into roots[]; returns count.
__device__ inline int poly_roots_real(const double* pin, int ncoef, double* roots, double* zr, double* zi, double* cm) {
  int lead = 0;
  while (lead + 1 < ncoef && pin[lead] == 0.0) ++lead;
  const double* p = pin + lead;
  const int deg = ncoef - lead - 1;
  if (deg <= 0) return 0;
  if (deg == 1) { roots[0] = -p[1] / p[0]; return 1; }
  if (deg == 2) {
    const double a = p[0], b = p[1], cc = p[2];
    const double D = b * b - 4 * a * cc;
    const double sD = sqrt(fabs(D));
    if (D >= 0) {
      if (b >= 0) { roots[0] = (-b - sD) / (2.0 * a); roots[1] = (2.0 * cc) / (-b - sD); }
      else { roots[0] = (2.0 * cc) / (-b + sD); roots[1] = (-b + sD) / (2.0 * a); }
    } else { roots[0] = -b / (2.0 * a); roots[1] = -b / (2.0 * a); }
    return 2;
  }
  // Aberth-Ehrlich on the monic polynomial
  double radius = 0.0;
  for (int i = 0; i <= deg; ++i) cm[i] = p[i] / p[0];
  for (int i = 1; i <= deg; ++i) radius = fmax(radius, pow(fabs(cm[i]), 1.0 / i));
  radius = fmax(2.0 * radius, 1e-300);
  for (int i = 0; i < deg; ++i) {
    double sn, cs;
    sincos(2.0 * M_PI * i / deg + 0.4, &sn, &cs);
    zr[i] = radius * cs; zi[i] = radius * sn;
  }
  for (int it = 0; it < 200; ++it) {
    double maxstep = 0.0;
    for (int i = 0; i < deg; ++i) {
      const double xr = zr[i], xi = zi[i];
      double pr = cm[0], pi = 0.0, dr = 0.0, di = 0.0;
      for (int kk = 1; kk <= deg; ++kk) {
        const double ndr = dr * xr - di * xi + pr, ndi = dr * xi + di * xr + pi;
        dr = ndr; di = ndi;
        const double npr = pr * xr - pi * xi + cm[kk], npi = pr * xi + pi * xr;
        pr = npr; pi = npi;
      }
      if (pr == 0.0 && pi == 0.0) continue;
      // ratio = p / p'
      const double dd = dr * dr + di * di;
      const double rr = (pr * dr + pi * di) / dd, ri = (pi * dr - pr * di) / dd;
      double sr = 0.0, si = 0.0;
      for (int j = 0; j < deg; ++j) {
        if (j == i) continue;
        const double er = xr - zr[j], ei = xi - zi[j];
        const double ee = er * er + ei * ei;
        sr += er / ee; si += -ei / ee;
      }
      // step = ratio / (1 - ratio * sum)
      const double qr = 1.0 - (rr * sr - ri * si), qi = -(rr * si + ri * sr);
      const double qq = qr * qr + qi * qi;
      const double str = (rr * qr + ri * qi) / qq, sti = (ri * qr - rr * qi) / qq;
      zr[i] = xr - str; zi[i] = xi - sti;
      const double mag = sqrt(zr[i] * zr[i] + zi[i] * zi[i]);
      maxstep = fmax(maxstep, sqrt(str * str + sti * sti) / fmax(1e-300, mag));
    }
    if (maxstep < 1e-15) break;
  }
  for (int i = 0; i < deg; ++i) roots[i] = zr[i];
  return deg;
}

struct Sample {
  double x, value, gradient;
  bool value_valid, gradient_valid;
};

// LineSearch::InterpolatingPolynomialMinimizingStepSize with CUBIC interpolation:
// fit a polynomial through {lowerbound, current[, previous]} (values and directional derivatives), minimise on
// [lo, hi]. scratch: >= 96 doubles of LDS.
__device__ __attribute__((noinline)) double interpolate_step(const Sample& lower, const Sample& previous, const Sample& current,
                                          double lo, double hi, double* scratch) {
  if (!current.value_valid) return fmin(fmax(current.x * 0.5, lo), hi);
  double* A = scratch;          // 36
  double* b = scratch + 36;     // 6
  double* z = scratch + 42;     // 6
  double* poly = scratch + 48;  // 6
  double* dpoly = scratch + 54; // 6
  double* roots = scratch + 60; // 6
  double* zr = scratch + 66;    // 6
  double* zi = scratch + 72;    // 6
  double* cm = scratch + 78;    // 6
  int* perm = (int*)(scratch + 84);  // 6 ints
  const bool use_prev = previous.value_valid;
  int nc = (lower.value_valid ? 1 : 0) + (lower.gradient_valid ? 1 : 0) + (current.value_valid ? 1 : 0) +
           (current.gradient_valid ? 1 : 0);
  if (use_prev) nc += (previous.value_valid ? 1 : 0) + (previous.gradient_valid ? 1 : 0);
  const int degree = nc - 1;
  int row = 0;
  auto add_sample = [&](const Sample& sm) {
    if (sm.value_valid) {
      double pw = 1.0;
      for (int j = degree; j >= 0; --j) { A[row * nc + j] = pw; pw *= sm.x; }
      b[row] = sm.value; ++row;
    }
    if (sm.gradient_valid) {
      double pw = 1.0;
      A[row * nc + degree] = 0.0;
      for (int j = degree - 1; j >= 0; --j) { A[row * nc + j] = (degree - j) * pw; pw *= sm.x; }
      b[row] = sm.gradient; ++row;
    }
  };
  add_sample(lower);
  add_sample(current);
  if (use_prev) add_sample(previous);
  fullpiv_solve(A, b, perm, z, poly, nc);
  // MinimizePolynomial on [lo, hi]
  double opt_x = (lo + hi) / 2.0;
  double opt_v = eval_poly(poly, nc, opt_x);
  const double vlo = eval_poly(poly, nc, lo);
  if (vlo < opt_v) { opt_v = vlo; opt_x = lo; }
  const double vhi = eval_poly(poly, nc, hi);
  if (vhi < opt_v) { opt_v = vhi; opt_x = hi; }
  if (nc > 2) {
    for (int i = 0; i < degree; ++i) dpoly[i] = (degree - i) * poly[i];
    const int nr = poly_roots_real(dpoly, degree, roots, zr, zi, cm);
    for (int i = 0; i < nr; ++i) {
      const double rt = roots[i];
      if (rt < lo || rt > hi) continue;
      const double v = eval_poly(poly, nc, rt);
      if (v < opt_v) { opt_v = v; opt_x = rt; }
    }
  }
  auto check_sample = [&](const Sample& sm) {
    if (sm.x < lo || sm.x > hi) return;
    const double v = eval_poly(poly, nc, sm.x);
    if (v < opt_v) { opt_x = sm.x; opt_v = v; }
  };
  check_sample(lower);
  check_sample(current);
  if (use_prev) check_sample(previous);
  return opt_x;
}

// ---- register-resident fast paths of the line-search interpolation (same algorithm as above, no LDS loops) ----

template <int nc> __device__ inline double eval_poly_reg(const double (&p)[nc], double x) {
  double v = 0.0;
#pragma unroll
  for (int i = 0; i < nc; ++i) v = v * x + p[i];
  return v;
}

// 1/x to ~1e-16 relative: hardware estimate + two Newton steps (only used where a last-bit error is harmless).
__device__ inline double fast_rcp(double x) {
  double r = __builtin_amdgcn_rcp(x);
  r = r * fma(-x, r, 2.0);
  r = r * fma(-x, r, 2.0);
  return r;
}

// sqrt(x) for x >= 0 (0 stays 0) through the refined reciprocal square root: 7 instructions instead of the library's
// ~20 (1-2 ulp; used for norms that only enter tolerance tests and for discriminants of the line search).
__device__ inline double fast_sqrt(double x) { return x > 0.0 ? x * rsqrt_pos(x) : 0.0; }

// Horner value and derivative of a degree-D polynomial p[0] x^D + ... + p[D].
template <int D> __device__ inline void horner_d(const double (&p)[D + 1], double x, double& f, double& df) {
  double v = p[0], d = 0.0;
#pragma unroll
  for (int i = 1; i <= D; ++i) { d = fma(d, x, v); v = fma(v, x, p[i]); }
  f = v; df = d;
}

// The root of p inside [a, b], a <= b, given that p is monotone there: Newton from the midpoint, kept inside the
// shrinking sign-change bracket (bisection whenever a Newton step leaves it). NaN if p does not change sign on [a, b].
template <int D> __device__ inline double bracketed_root(const double (&p)[D + 1], double a, double b, int* trips = nullptr) {
  double fa, fb, t;
  horner_d<D>(p, a, fa, t);
  horner_d<D>(p, b, fb, t);
  if (fa == 0.0) return a;
  if (fb == 0.0) return b;
  if (!((fa < 0.0) != (fb < 0.0)) || !(fa == fa) || !(fb == fb)) return __builtin_nan("");
  const bool neg_lo = fa < 0.0;
  // start from the secant point of the bracket (inside it, since the signs differ); Newton from there, bisection
  // whenever a step leaves the shrinking bracket
  double xl = a, xh = b, x = a - fa * (b - a) * fast_rcp(fb - fa);
  if (!(x > a && x < b)) x = 0.5 * (a + b);
  for (int it = 0; it < 80; ++it) {
    SMPC_LS_COUNT(D == 4 ? 1 : 3, 1);
    if (trips) ++*trips;
    double fx, dfx;
    horner_d<D>(p, x, fx, dfx);
    if (fx == 0.0) break;
    if ((fx < 0.0) == neg_lo) xl = x; else xh = x;
    double xn = x - fx * fast_rcp(dfx);
    // inclusive: a converged iterate moves by less than an ulp, lands ON the bracket end it has just become, and must
    // count as a (zero-length) Newton step — with strict inequalities it was sent back to the midpoint of a still wide
    // bracket and the search started over (measured on real line searches: 11 % of the calls took 16..58 trips)
    const bool newton = xn >= xl && xn <= xh;
    if (!newton) xn = 0.5 * (xl + xh);
    // A Newton step below 1e-9 |x| leaves an error of the order of its square; waiting for the step itself to reach
    // round-off would spin on polynomials whose Horner value is noisier than that (measured: 7% of the solve kernel).
    const double dx = fabs(xn - x);
    const bool done = (newton && dx <= 1e-9 * fabs(xn)) || dx <= 4e-16 * fabs(xn);
    x = xn;
    if (done) break;
  }
  SMPC_LS_COUNT(D == 4 ? 0 : 2, 1);
  return x;
}

// Real roots of the quartic q (q[0] != 0) inside [lo, hi] — the only roots of the derivative that can win the
// minimisation of the interpolating quintic (the real part of a complex pair, which the reference's companion-matrix /
// this repository's Aberth fallback also offer as candidates, is never a critical point and so never below the minimum
// over {lo, hi, real critical points}). Isolation by monotone pieces: the roots of q'' (quadratic, closed form) cut
// [lo, hi] into <= 3 pieces on which q' is monotone; its <= 3 roots there cut [lo, hi] into <= 4 pieces on which q is
// monotone. Four neighbouring lanes take one piece each (lane & 3); roots[j] is NaN where piece j holds no root.
__device__ inline void quartic_roots_in_range_lanes(const double (&q)[5], double lo, double hi, double (&roots)[4]) {
  int lane = threadIdx.x & 63;
  asm volatile("" : "+v"(lane));  // (the masks "r == j" stay here: hoisted out of the persistent loop they were spilled SGPR pairs)
  const int r = lane & 3, base = lane & ~3;
  const double A = 12.0 * q[0], Bq = 6.0 * q[1], C = 2.0 * q[2];
  double e0 = lo, e1 = lo;
  bool inflection_inside = false;
  const double D = Bq * Bq - 4.0 * A * C;
  if (D > 0.0) {
    const double t = -0.5 * (Bq + copysign(fast_sqrt(D), Bq));
    const double x1 = t * fast_rcp(A), x2 = C * fast_rcp(t);
    inflection_inside = (x1 > lo && x1 < hi) || (x2 > lo && x2 < hi);
    e0 = fmin(fmax(fmin(x1, x2), lo), hi);
    e1 = fmax(fmin(fmax(x1, x2), hi), lo);
  }
  {
    // The common shape of a line-search interpolant (96 % of 6850 quintic fits dumped from real solves): q = p' has
    // opposite signs at the two ends and no inflection in between (q'' keeps its sign, q is convex or concave), so q
    // has exactly one root there — the only critical point of p in the interval. One bracketed iteration, the same in
    // every lane, instead of the two lane-parallel isolation stages below; and when q falls through zero the point is
    // a local maximum of p, which can never win against the interval ends: nothing to compute at all.
    double ql, qh, t;
    horner_d<4>(q, lo, ql, t);
    horner_d<4>(q, hi, qh, t);
    const bool one_root = !inflection_inside && ((ql < 0.0 && qh > 0.0) || (ql > 0.0 && qh < 0.0));
    if (one_root) SMPC_LS_COUNT(6, 1);
    if (one_root) {  // decided per slot (the value is the same in all its lanes): a result never depends on the wave's other scene
      roots[0] = (ql < 0.0) ? bracketed_root<4>(q, lo, hi) : __builtin_nan("");
      roots[1] = roots[2] = roots[3] = __builtin_nan("");
      return;
    }
  }
  const double d1[4] = {4.0 * q[0], 3.0 * q[1], 2.0 * q[2], q[3]};
  const double pa = (r == 0) ? lo : (r == 1) ? e0 : e1;
  const double pb = (r == 0) ? e0 : (r == 1) ? e1 : hi;
  const double s = bracketed_root<3>(d1, pa, pb);
  const double s0 = __shfl(s, base + 0, 64), s1 = __shfl(s, base + 1, 64), s2 = __shfl(s, base + 2, 64);
  const double b1 = (s0 == s0) ? s0 : lo;
  const double b2 = (s1 == s1) ? fmax(s1, b1) : b1;
  const double b3 = (s2 == s2) ? fmax(s2, b2) : b2;
  const double ca = (r == 0) ? lo : (r == 1) ? b1 : (r == 2) ? b2 : b3;
  const double cb = (r == 0) ? b1 : (r == 1) ? b2 : (r == 2) ? b3 : hi;
  const double root = bracketed_root<4>(q, ca, cb);
#pragma unroll
  for (int j = 0; j < 4; ++j) roots[j] = __shfl(root, base + j, 64);
}

// MinimizeInterpolatingPolynomial for the two common shapes: {lower, current} with all values and gradients valid
// (cubic) and {lower, current, previous} (quintic). The lower bound sample always sits at x = 0, so its two
// interpolation conditions fix the two lowest coefficients exactly (p(0) = f0, p'(0) = g0; in the full-pivot LU of
// FindInterpolatingPolynomial those two unit rows are never touched by an elimination step either); the remaining
// 2 / 4 coefficients come from the reduced system of the other samples. Returns false if the shape is not covered.
__device__ inline bool interpolate_step_fast(const Sample& lower, const Sample& previous, const Sample& current,
                                             double lo, double hi, double& step_size) {
  if (!(lower.value_valid && lower.gradient_valid && current.value_valid && current.gradient_valid)) return false;
  if (lower.x != 0.0) return false;
  const bool use_prev = previous.value_valid;
  if (use_prev && !previous.gradient_valid) return false;
  const double f0 = lower.value, g0 = lower.gradient;
  double opt_x = (lo + hi) / 2.0, opt_v;
  SMPC_LS_COUNT(use_prev ? 5 : 4, 1);
  if (!use_prev) {
    constexpr int nc = 4;
    // a x1^3 + b x1^2 = f1 - g0 x1 - f0 =: A ;  3 a x1^2 + 2 b x1 = g1 - g0 =: B, solved in closed form (the reference
    // runs a full-pivot LU on the 4 x 4 Vandermonde system: the same polynomial up to its round-off)
    const double x1 = current.x;
    if (x1 == 0.0) return false;
    const double rx = fast_rcp(x1);
    const double A = current.value - g0 * x1 - f0, Bx = (current.gradient - g0) * x1;
    const double rxsq = rx * rx;
    const double poly[nc] = {(Bx - 2.0 * A) * (rxsq * rx), (3.0 * A - Bx) * rxsq, g0, f0};
    opt_v = eval_poly_reg<nc>(poly, opt_x);
    const double vlo = eval_poly_reg<nc>(poly, lo);
    if (vlo < opt_v) { opt_v = vlo; opt_x = lo; }
    const double vhi = eval_poly_reg<nc>(poly, hi);
    if (vhi < opt_v) { opt_v = vhi; opt_x = hi; }
    // derivative 3 p0 x^2 + 2 p1 x + p2, roots as poly_roots_real() finds them (leading zeros stripped)
    const double qa = 3.0 * poly[0], qb = 2.0 * poly[1], qc = poly[2];
    double r0 = 0.0, r1 = 0.0;
    int nr = 0;
    if (qa != 0.0) {
      const double D = qb * qb - 4 * qa * qc;
      const double sD = fast_sqrt(fabs(D));
      const double inv2a = fast_rcp(2.0 * qa);
      if (D >= 0) {
        // the stable pair of formulas: tq = -(qb + sign(qb) sD) is the sum without cancellation
        const double tq = (qb >= 0) ? (-qb - sD) : (-qb + sD);
        const double big = tq * inv2a, small = (2.0 * qc) * fast_rcp(tq);
        r0 = (qb >= 0) ? big : small;
        r1 = (qb >= 0) ? small : big;
      } else { r0 = -qb * inv2a; r1 = r0; }
      nr = 2;
    } else if (qb != 0.0) { r0 = -qc * fast_rcp(qb); nr = 1; }
    // Candidates: the critical points and the samples themselves, each only if it lies in [lo, hi]. In a backtracking
    // search the interval is [1e-3, 0.6] x the current step, so the samples (0, the current step, the longer previous
    // one) never do, and of the two critical points at most the minimum: every evaluation sits behind a wave-level
    // test (results unchanged: a candidate outside the interval was never taken).
    // (x >= lo && x <= hi rather than the reference's !(x < lo || x > hi): a NaN candidate, which the reference evaluates
    // to a NaN value that never wins, is simply not evaluated)
    const bool in0 = nr >= 1 && r0 >= lo && r0 <= hi, in1 = nr >= 2 && r1 >= lo && r1 <= hi;
    if (__any(in0)) { const double v = eval_poly_reg<nc>(poly, r0); if (in0 && v < opt_v) { opt_v = v; opt_x = r0; } }
    if (__any(in1)) { const double v = eval_poly_reg<nc>(poly, r1); if (in1 && v < opt_v) { opt_v = v; opt_x = r1; } }
    const bool inl = lower.x >= lo && lower.x <= hi, inc = current.x >= lo && current.x <= hi;
    if (__any(inl)) { const double v = eval_poly_reg<nc>(poly, lower.x); if (inl && v < opt_v) { opt_v = v; opt_x = lower.x; } }
    if (__any(inc)) { const double v = eval_poly_reg<nc>(poly, current.x); if (inc && v < opt_v) { opt_v = v; opt_x = current.x; } }
    step_size = opt_x;
    return true;
  }
  constexpr int nc = 6;
  // Quintic through {0: f0, g0; x1 = current; x2 = previous}: Hermite divided differences on the nodes 0, 0, x1, x1, x2, x2
  // and expansion of the Newton form. The reference solves the 6 x 6 Vandermonde system by full-pivot LU; against an
  // exact solve the divided differences are the more accurate of the two (argmin within 3e-15 vs 8e-12 relative,
  // tools/ — measured on 3000 random line searches), and they cost ~60 instructions instead of ~700 of select-based
  // pivoting.
  const double x1 = current.x, x2 = previous.x;
  if (x1 == 0.0 || x2 == 0.0 || x1 == x2) return false;
  const double r1 = fast_rcp(x1), r2 = fast_rcp(x2), r12 = fast_rcp(x2 - x1);
  const double d01 = (current.value - f0) * r1, d12 = (previous.value - current.value) * r12;
  const double e0 = (d01 - g0) * r1, e1 = (current.gradient - d01) * r1;
  const double e2 = (d12 - current.gradient) * r12, e3 = (previous.gradient - d12) * r12;
  const double h0 = (e1 - e0) * r1, h1 = (e2 - e1) * r2, h2 = (e3 - e2) * r12;
  const double k0 = (h1 - h0) * r2, k1 = (h2 - h1) * r2;
  const double m0 = (k1 - k0) * r2;
  const double x1s = x1 * x1;
  const double poly[nc] = {m0, k0 - m0 * (2.0 * x1 + x2), h0 - 2.0 * k0 * x1 + m0 * (x1s + 2.0 * x1 * x2),
                           e0 - h0 * x1 + k0 * x1s - m0 * x1s * x2, g0, f0};
  double dq[5];
#pragma unroll
  for (int i = 0; i < 5; ++i) dq[i] = (5 - i) * poly[i];
  if (dq[0] == 0.0) return false;  // degenerate leading coefficient: generic path
  opt_v = eval_poly_reg<nc>(poly, opt_x);
  const double vlo = eval_poly_reg<nc>(poly, lo);
  if (vlo < opt_v) { opt_v = vlo; opt_x = lo; }
  const double vhi = eval_poly_reg<nc>(poly, hi);
  if (vhi < opt_v) { opt_v = vhi; opt_x = hi; }
  double roots[4];
  quartic_roots_in_range_lanes(dq, lo, hi, roots);
  // (each candidate behind a wave-level test, as in the cubic case: roots[1..3] are NaN after the one-root shortcut)
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const double rt = roots[i];
    const bool in = rt >= lo && rt <= hi;
    if (__any(in)) { const double v = eval_poly_reg<nc>(poly, rt); if (in && v < opt_v) { opt_v = v; opt_x = rt; } }
  }
  const double sx[3] = {lower.x, current.x, previous.x};
#pragma unroll
  for (int smp = 0; smp < 3; ++smp) {
    const bool in = sx[smp] >= lo && sx[smp] <= hi;
    if (__any(in)) { const double v = eval_poly_reg<nc>(poly, sx[smp]); if (in && v < opt_v) { opt_v = v; opt_x = sx[smp]; } }
  }
  step_size = opt_x;
  return true;
}

// std::min(std::max(v, lo), hi) of parameter_block.h Plus(): a NaN stays a NaN (fmin / fmax alone would drop it and a
// NaN warm start would be solved from the lower bound instead of failing its initial evaluation like Ceres)
__device__ inline double clampd(double v, double lo, double hi) { return (v != v) ? v : fmin(fmax(v, lo), hi); }

// tf2 Quaternion::setRPY(0,0,yaw) followed by tf2::getYaw (x = y = 0): src/optimizer.cpp:434-439 round trips.
__device__ inline double yaw_roundtrip(double yaw) {
  double sz, cz;
  sincos(yaw * 0.5, &sz, &cz);
  return atan2(2.0 * (cz * sz), cz * cz - sz * sz);
}


// Armijo sufficient decrease of a line-search sample, the one expression both the sweep (does the
// sample need its whole Gram?) and the state machine (is the search over?) evaluate: explicit operations, so that the two
// sites cannot be contracted differently. A NaN value fails.
__device__ inline bool armijo_holds(double value, double cost, double gd0, double alpha) {
  return isfinite(value) && !(value > fma(1e-4 * gd0, alpha, cost));
}

enum Phase { PH_FETCH = 0, PH_INIT = 1, PH_LS = 2, PH_REEVAL = 3, PH_DONE = 4, PH_IDLE = 5 };

// Slot-uniform LM scalars parked in LDS (offsets into the scal[] block).
enum Scal { S_COST = 0, S_XNORM, S_GMAX, S_RADIUS, S_DECF, S_MCC, S_GD0, S_DIRMAX, S_PREV_X, S_PREV_V, S_PREV_G,
            S_CUR_X, S_CUR_V, S_CUR_G, S_INITIAL_COST, S_FIRST_V, S_COUNT };

struct LmRegs {  // slot-uniform integers / flags kept in registers
  int phase, iter, evals, num_invalid, ls_iters, n_samples, status, reason;
  bool step_successful, at_least_one, prev_vv, prev_gv, cur_vv, cur_gv, first_vv;
};

// waves per SIMD the solve kernel's register allocation must allow: three for the two-scenes-per-wave kernels up to three
// parameter blocks (the headline shapes sit at 160-168 registers; stated so that an edit cannot silently cost the third
// wave, which is what overlapped launches of several streams live on), two otherwise — the one-scene-per-wave kernels
// serve small batches (the plugin's own B = 1 call) and long horizons, whose launches take at most eight waves per CU,
// and their helper-lane loop needs the registers (held to 168 it spilled 14)
#ifndef SMPC_SOLVE_MIN_WAVES
#define SMPC_SOLVE_MIN_WAVES(NB, W) (((NB) <= 3 && (W) == 32) ? 3 : 2)
#endif
// The LM vectors and matrices of a slot are spread over its lanes: lane q < P owns parameter q (its entry of x, of the
// trial point, of the step, row q of the scaled Gram and of its Cholesky factor). One instruction then updates all P
// entries; sums over the parameters go through a few LDS words in index order (the same order a serial loop would
// add them in). Nothing P x P lives in registers, so the P = 8..12 instantiations do not spill.
template <int NB, int W, bool kVT = false>
__global__ __launch_bounds__(64, SMPC_SOLVE_MIN_WAVES(NB, W)) void smpc_solve_kernel(const KParams) {
  const auto& k = *(KParamsK)__builtin_amdgcn_kernarg_segment_ptr();
  constexpr int P = 2 * NB;
  constexpr int S = kWave / W;
  extern __shared__ __attribute__((aligned(32))) double lds_all[];
  const int lane = threadIdx.x & 63;
  Ctx c;
  c.kp = &k;
  c.L = make_layout(k.T, k.N, P, kLayoutSolve, W);
  c.ag = k.people_rec;
  {
    double* atab = lds_all + atan_tab_offset(S * c.L.total, wave_extra_doubles(P, W));
    load_atan_nodes(c.kp, atab, lane);
    c.atab = atab;
  }
  const auto& prm = k.prm;
  const int T = k.T;
  // Everything below is derived from the lane index. It is re-derived at the top of every trip and again behind the
  // sweep from a copy of the lane index the compiler cannot see through (an empty asm): otherwise these ~20 addresses
  // and flags are computed once in the prologue, stay live through the whole kernel — the sweep runs at the VGPR limit —
  // and come back as scratch reloads inside the loop.
  int slot, q, qc;
  bool act;
  double *Hs, *Lw, *gs, *gu, *xc, *xt, *dl, *sc, *bc, *rs, *sv, *scratch;
  int32_t* rw;
  auto bind = [&](int lane_t) {
    slot = lane_t / W;
    c.sl = lane_t - slot * W;
    c.slot = slot;
    c.lds = lds_all + (size_t)slot * c.L.total;
    c.wave_lds = lds_all + (size_t)S * c.L.total;
    Hs = c.lds + c.L.lm;   // [P][P] scaled J^T J at the current point, dense
    gs = Hs + P * P;       // [P] scaled gradient
    gu = gs + P;           // [P] unscaled gradient
    xc = gu + P;           // [P] current point
    xt = xc + P;           // [P] trial point (input of the sweep)
    dl = xt + P;           // [P] delta (unscaled step of this iteration)
    sc = dl + P;           // [P] Jacobi scaling
    sv = sc + P;           // scalars [24]
    rw = reinterpret_cast<int32_t*>(sv + S_COUNT);  // the state machine's integers, parked across the sweep
    // temporaries of the LM algebra: over the sweep's cos / sin block, scans and Gram reduction buffer (all dead here)
    Lw = c.lds + c.L.cs;   // [P][P] rows of the Cholesky factor of the damped system
    bc = Lw + P * P;       // [4][P] hand-over words: pivots, forward / backward solutions, scaled step
    rs = bc + 4 * P;       // [3][P] reduction words (every site uses the same three rows: LDS operations of a wave
                           //        execute in program order, a site's reads are behind it before the next site writes)
    scratch = rs + 3 * P;  // [96] generic line-search interpolation fallback
    q = c.sl;              // the parameter this lane owns in the LM algebra
    act = q < P;
    qc = act ? q : 0;      // in-range index for lanes that only tag along
  };
  bind(lane);
  auto slot_any = [&](bool pred) -> bool {
    const unsigned long long slot_bits = (W == 64) ? ~0ull : (0xffffffffull << (32 * slot));
    return (__ballot(pred) & slot_bits) != 0ull;
  };
  // sums / maximum over the parameters: every active lane leaves its terms in the site's words, then every lane adds
  // them up in index order
  auto reduce3 = [&](int site, double a, double b, double m, double& sa, double& sb, double& sm) {
    double* w3 = rs;
    (void)site;
    if (act) { w3[q] = a; w3[P + q] = b; w3[2 * P + q] = m; }
    wave_lds_fence();
    sa = 0.0; sb = 0.0; sm = 0.0;
#pragma unroll
    for (int i = 0; i < P; ++i) { sa += w3[i]; sb += w3[P + i]; sm = fmax(sm, w3[2 * P + i]); }
  };

  // The state machine's integers and flags live in LDS across the sweep (behind the scalars of sv[]): only the phase
  // stays in a register there. The sweep is the register-hungry part of a trip; nothing of the LM bookkeeping should
  // take room in it.
  LmRegs R;
  R.phase = PH_FETCH;
  R.iter = R.evals = R.num_invalid = R.ls_iters = R.n_samples = 0;
  R.status = SMPC_NO_CONVERGENCE; R.reason = SMPC_REASON_MAX_ITERATIONS;
  R.step_successful = R.at_least_one = R.prev_vv = R.prev_gv = R.cur_vv = R.cur_gv = R.first_vv = false;
  auto park = [&]() {
    rw[0] = R.iter; rw[1] = R.evals; rw[2] = R.num_invalid; rw[3] = R.ls_iters; rw[4] = R.n_samples; rw[5] = R.status;
    rw[6] = R.reason;
    rw[7] = (R.step_successful ? 1 : 0) | (R.at_least_one ? 2 : 0) | (R.prev_vv ? 4 : 0) | (R.prev_gv ? 8 : 0) |
            (R.cur_vv ? 16 : 0) | (R.cur_gv ? 32 : 0) | (R.first_vv ? 64 : 0);
  };
  auto unpark = [&]() {
    R.iter = rw[0]; R.evals = rw[1]; R.num_invalid = rw[2]; R.ls_iters = rw[3]; R.n_samples = rw[4]; R.status = rw[5];
    R.reason = rw[6];
    const int f = rw[7];
    R.step_successful = f & 1; R.at_least_one = f & 2; R.prev_vv = f & 4; R.prev_gv = f & 8;
    R.cur_vv = f & 16; R.cur_gv = f & 32; R.first_vv = f & 64;
  };
  bool ever_loaded = false;
#ifdef SMPC_STAMPS
  for (int i = 0; i < 8; ++i) c.acc[i] = 0;
  for (int i = 0; i < 4; ++i) c.acc2[i] = 0;
  c.t_last = __builtin_amdgcn_s_memtime();
#endif

  for (;;) {
    SMPC_STAMP(c, 6);  // LM state machine + output stage of the previous trip
    // Like everything derived from the lane index (above), everything derived from the launch parameters is re-derived
    // per trip, behind an opaque copy of the argument pointer: sign extensions of T and N, "N > 1", the LDS layout's
    // offsets, ... computed once in the prologue are ~20 scalar registers live through the whole kernel — more than the
    // file has left; they came back as v_writelane / v_readlane spill traffic (22 spilled SGPRs in <3,32>, 70 in <5,64>;
    // now 4 and 33). A few dozen scalar instructions and cached scalar loads per trip.
    {
      KParamsK kp_t = c.kp;
      asm volatile("" : "+s"(kp_t));
      c.kp = kp_t;
    }
    const auto& k = *c.kp;
    const auto& prm = k.prm;
    const int T = k.T;
    c.L = make_layout(k.T, k.N, P, kLayoutSolve, W);
    {
      int lane_t = lane;
      asm volatile("" : "+v"(lane_t));
      bind(lane_t);
    }
    // ---------------------------------------------------------------- fetch the next scene for idle slots
    if (R.phase == PH_FETCH) {
      int scene = 0;
      if (c.sl == 0) {
        scene = atomicAdd(k.queue, 1);
        if (k.order) {  // the caller's order (longest scenes first); an entry outside the batch is skipped
          while (scene < k.B) {
            const int want = k.order[scene];
            if (want >= 0 && want < k.B) { scene = want; break; }
            scene = atomicAdd(k.queue, 1);
          }
        }
      }
      scene = __shfl(scene, slot * W, 64);
      if (scene < k.B) {
        load_scene<W, kVT>(c, scene);
        ever_loaded = true;
        wave_lds_fence();
        const Horizon hz0 = get_horizon<NB, kVT>(c);
        double v = 0.0;
        if (act && (q >> 1) <= hz0.blast) {  // a scene with fewer blocks than NB keeps the surplus parameters at zero
          const bool bnd = (q >> 1) < hz0.nbounded;
          const double lo0 = bnd ? ((q & 1) ? prm.w_min : prm.v_min) : -1.7976931348623157e308;
          const double hi0 = bnd ? ((q & 1) ? prm.w_max : prm.v_max) : 1.7976931348623157e308;
          v = clampd(k.init_params[(size_t)scene * P + q] + 0.0, lo0, hi0);  // Plus(x, 0): project the start point (A.4)
        }
        if (act) { xc[q] = v; xt[q] = v; }
        double xn, u0, u1;
        reduce3(0, v * v, 0.0, 0.0, xn, u0, u1);
        sv[S_XNORM] = fast_sqrt(xn);
        R.phase = PH_INIT;
        R.iter = 0; R.evals = 0; R.num_invalid = 0;
        R.status = SMPC_NO_CONVERGENCE; R.reason = SMPC_REASON_MAX_ITERATIONS;
        R.step_successful = true; R.at_least_one = false;
      } else {
        if (!ever_loaded) {  // keep the sweep's memory accesses in bounds for a slot that never got a scene
          load_scene<W, kVT>(c, 0);
          ever_loaded = true;
          if (act) xt[q] = 0.0;
        }
        R.phase = PH_IDLE;
      }
    }
    if (__all(R.phase == PH_IDLE)) break;
    SMPC_STAMP(c, 0);  // fetch + load_scene
    if (k.prio_step > 0) {
      // Attained-service priority: the launch ends when its longest scene does, and a scene's sweeps are a dependent
      // chain — so the longer a scene has been running, the more of the SIMD's issue slots its wave gets against the
      // younger waves beside it (which, having made few sweeps, most likely hold short scenes: lengths 11..146).
      const int mine = (R.phase == PH_IDLE) ? 0 : R.evals;
      int age = __builtin_amdgcn_readfirstlane(mine);
      if (S == 2) age = max(age, __builtin_amdgcn_readlane(mine, 32));
      const int step = k.prio_step;
      if (age >= 3 * step) __builtin_amdgcn_s_setprio(3);
      else if (age >= 2 * step) __builtin_amdgcn_s_setprio(2);
      else if (age >= step) __builtin_amdgcn_s_setprio(1);
      else __builtin_amdgcn_s_setprio(0);
    }

    // ---------------------------------------------------------------- one sweep for every slot of the wave
    park();
    // What the slot needs of this sweep: the whole Gram where the point can be adopted (the initial point, a sample that
    // passes the Armijo test, a re-evaluation), its last column — cost and gradient — where the sample only feeds the
    // line search's interpolation (the common case: 19 of a solve's 52 sweeps are adopted on the headline workload).
    const int phase_at_sweep = R.phase;
    auto need_rest = [&]() -> bool {
      if (phase_at_sweep != PH_LS) return phase_at_sweep != PH_IDLE;
      if (k.full_gram) return true;
      const double* gt = c.lds + c.L.gram;
      const double* svp = c.lds + c.L.lm + P * P + 6 * P;
      const unsigned long long slot_bits = (W == 64) ? ~0ull : (0xffffffffull << (32 * c.slot));
      // a non-finite residual or Jacobian entry shows in this column too (a product with it is not finite): only then,
      // and then always, the diagonal is formed as well and decides as before
      const bool col_finite = (__ballot(c.sl <= P && !isfinite(gt[min(c.sl, P) * (P + 1) + P])) & slot_bits) == 0ull;
      return !col_finite || armijo_holds(0.5 * gt[P * (P + 1) + P], svp[S_COST], svp[S_GD0], svp[S_CUR_X]);
    };
    sweep<NB, W, false, kVT>(c, xt, nullptr, nullptr, need_rest);  // [J r]^T [J r] of this slot, left in LDS
    {
      int lane_t = lane;
      asm volatile("" : "+v"(lane_t));
      bind(lane_t);
    }
    GramView GH;
    GH.base = c.lds + c.L.gram;
    GH.ld = P + 1;
    unpark();
    const Horizon hz = get_horizon<NB, kVT>(c);
    // bounds of parameter q (src/optimizer.cpp:373-379: blocks 0..CH/bl-1 are bounded)
    const bool bounded = act && (q >> 1) < hz.nbounded;
    const double lo_q = bounded ? ((q & 1) ? prm.w_min : prm.v_min) : -1.7976931348623157e308;
    const double hi_q = bounded ? ((q & 1) ? prm.w_max : prm.v_max) : 1.7976931348623157e308;
    // usable iff every residual and Jacobian entry was finite: a non-finite one makes its diagonal Gram entry non-finite
    // (a sweep that stopped at the last column had every entry of that column finite, see need_rest)
    const bool gram_full = c.gram_full;
    const bool finite = !gram_full || !slot_any(c.sl <= P && !isfinite(GH.base[min(c.sl, P) * GH.ld + min(c.sl, P)]));
    const double val = 0.5 * GH(P, P);
    bool new_iteration = false;

    // ---------------------------------------------------------------- advance the slot's state machine
    auto adopt_trial_point = [&]() {  // x <- xt, Hs / gs / gu / gmax from G (scaled by the fixed Jacobi scaling)
      double xa = 0.0, gm = 0.0;
      if (act) {
        xa = xt[q];
        xc[q] = xa;
        const double scq = sc[q];
        const double* grow = GH.base + q * GH.ld;
#pragma unroll
        for (int b = 0; b < P; ++b) Hs[q * P + b] = grow[b] * scq * sc[b];  // the Gram is bitwise symmetric
        const double g = grow[P];
        gu[q] = g; gs[q] = g * scq;
        gm = fabs(xa - clampd(xa - g, lo_q, hi_q));
      }
      double xn, u0, gmax;
      reduce3(1, xa * xa, 0.0, gm, xn, u0, gmax);
      sv[S_XNORM] = fast_sqrt(xn);
      sv[S_GMAX] = gmax;
    };
    auto candidate = [&]() {  // A.9 tests on the candidate = current trial point; A.10 strategy update
      const double cost = sv[S_COST];
      const double cand_cost = R.cur_vv ? sv[S_CUR_V] : 1.7976931348623157e308;
      const double d = act ? xc[q] - xt[q] : 0.0;
      double sn2, u0, u1;
      reduce3(2, d * d, 0.0, 0.0, sn2, u0, u1);
      const double step_norm = fast_sqrt(sn2);
      const bool tol_allowed = !prm.fixed_iterations && (!prm.tol_needs_successful_step || R.at_least_one);
      if (tol_allowed && step_norm <= prm.param_tol * (sv[S_XNORM] + prm.param_tol)) {
        R.status = SMPC_CONVERGENCE; R.reason = SMPC_REASON_PARAMETER_TOL; R.phase = PH_DONE; return;
      }
      const double cost_change = cost - cand_cost;
      if (tol_allowed && fabs(cost_change) <= prm.fn_tol * cost) {
        R.status = SMPC_CONVERGENCE; R.reason = SMPC_REASON_FUNCTION_TOL; R.phase = PH_DONE; return;
      }
      const double rho = (cand_cost >= 1.7976931348623157e308) ? -1.7976931348623157e308 : div_fast(cost_change, sv[S_MCC]);
      if (rho > 1e-3 && !gram_full) {
        // never seen: a candidate that failed the Armijo test cannot be accepted (cost - value < -1e-4 g.delta <
        // 1e-3 model_cost_change). Should rounding ever say otherwise, the point is swept again with its whole Gram.
        R.phase = PH_REEVAL;
        return;
      }
      if (rho > 1e-3) {
        adopt_trial_point();
        sv[S_COST] = cand_cost;
        R.step_successful = true; R.at_least_one = true;
        const double t = 2.0 * rho - 1.0;
        sv[S_RADIUS] = fmin(1e16, div_fast(sv[S_RADIUS], fmax(1.0 / 3.0, 1.0 - t * t * t)));
        sv[S_DECF] = 2.0;
      } else {
        sv[S_RADIUS] = sv[S_RADIUS] / sv[S_DECF];
        sv[S_DECF] *= 2.0;
      }
      new_iteration = true;
    };

    if (R.phase == PH_INIT) {
      ++R.evals;
      sv[S_COST] = val;
      sv[S_INITIAL_COST] = val;
      if (kVT && reinterpret_cast<const int*>(c.lds + c.L.hz)[6] != 0) {
        R.status = SMPC_FAILURE; R.reason = SMPC_REASON_SHORT_PATH; R.phase = PH_DONE;  // "Path has less than 2 points"
      } else if (!finite) {
        R.status = SMPC_FAILURE; R.reason = SMPC_REASON_EVAL_FAILED; R.phase = PH_DONE;
      } else {
        if (act) sc[q] = 1.0 / (1.0 + sqrt(GH.base[q * GH.ld + q]));  // Jacobi scaling (A.5)
        wave_lds_fence();
        adopt_trial_point();
        sv[S_RADIUS] = 1e4; sv[S_DECF] = 2.0;
        new_iteration = true;
      }
    } else if (R.phase == PH_LS) {
      ++R.evals;
      // record the sample just evaluated (LineSearchFunction::Evaluate, A.8)
      const double dq = act ? dl[q] : 0.0;
      double gd, u0, u1;
      reduce3(3, act ? dq * GH.base[q * GH.ld + P] : 0.0, 0.0, 0.0, gd, u0, u1);
      R.cur_vv = finite && isfinite(val);
      R.cur_gv = R.cur_vv && isfinite(gd);
      sv[S_CUR_V] = val; sv[S_CUR_G] = gd;
      const double alpha = sv[S_CUR_X];
      if (R.n_samples == 1) { sv[S_FIRST_V] = val; R.first_vv = R.cur_vv; }  // the full step: candidate if the search fails
      if (R.cur_vv && armijo_holds(val, sv[S_COST], sv[S_GD0], alpha)) {
        // Armijo satisfied: delta *= alpha; the candidate is this very point
        if (act) dl[q] = dq * alpha;
        candidate();
      } else {
        ++R.ls_iters;
        bool failed = R.ls_iters >= 20;
        double step_size = 0.0;
        if (!failed) {
          Sample lower{0.0, sv[S_COST], sv[S_GD0], true, true};
          Sample previous{sv[S_PREV_X], sv[S_PREV_V], sv[S_PREV_G], R.prev_vv, R.prev_gv};
          Sample current{alpha, val, gd, R.cur_vv, R.cur_gv};
          SMPC_STAMP(c, 6);
          if (!interpolate_step_fast(lower, previous, current, 1e-3 * alpha, 0.6 * alpha, step_size))
          { SMPC_LS_COUNT(7, 1); step_size = interpolate_step(lower, previous, current, 1e-3 * alpha, 0.6 * alpha, scratch); }
          SMPC_STAMP(c, 7);  // line-search interpolation
          failed = step_size * sv[S_DIRMAX] < 1e-9;
        }
        if (!failed) {
          sv[S_PREV_X] = alpha; sv[S_PREV_V] = val; sv[S_PREV_G] = gd; R.prev_vv = R.cur_vv; R.prev_gv = R.cur_gv;
          sv[S_CUR_X] = step_size;
          if (act) xt[q] = clampd(xc[q] + step_size * dq, lo_q, hi_q);
          ++R.n_samples;
        } else if (R.n_samples > 1) {
          // Line search failed: delta unchanged, the candidate is the full step again (the first sample). Its cost is
          // known; its Gram is only needed if the step were accepted, which a step that failed the Armijo test at
          // alpha = 1 cannot be (cost - value_1 < -1e-4 g.delta < 1e-3 model_cost_change). Only in that never-seen case
          // the point is swept again (PH_REEVAL) so that the accepted state is built from its own Gram.
          if (act) xt[q] = clampd(xc[q] + dq, lo_q, hi_q);
          const double v1 = R.first_vv ? sv[S_FIRST_V] : 1.7976931348623157e308;
          const bool would_accept = R.first_vv && ((sv[S_COST] - v1) / sv[S_MCC] > 1e-3);
          if (would_accept) {
            R.phase = PH_REEVAL;
          } else {
            R.cur_vv = R.first_vv;
            sv[S_CUR_V] = v1;
            wave_lds_fence();
            candidate();
          }
        } else {
          candidate();  // the only sample was the full step itself
        }
      }
    } else if (R.phase == PH_REEVAL) {
      ++R.evals;
      R.cur_vv = finite && isfinite(val);
      sv[S_CUR_V] = val;
      candidate();
    }

    // ---------------------------------------------------------------- start the next LM iteration (A.6, A.7)
    if (new_iteration) {
      for (;;) {
        if (R.iter >= prm.max_iterations) { R.status = SMPC_NO_CONVERGENCE; R.reason = SMPC_REASON_MAX_ITERATIONS; R.phase = PH_DONE; break; }
        if (R.step_successful && sv[S_GMAX] <= prm.gradient_tol && !prm.fixed_iterations) { R.status = SMPC_CONVERGENCE; R.reason = SMPC_REASON_GRADIENT_TOL; R.phase = PH_DONE; break; }
        if (sv[S_RADIUS] <= 1e-32) { R.status = SMPC_CONVERGENCE; R.reason = SMPC_REASON_MIN_RADIUS; R.phase = PH_DONE; break; }
        ++R.iter;
        R.step_successful = false;
        // (the column masks "q == j" / "q > j" of this block stay inside it: hoisted out of this retry loop they were two
        // scalar registers each, 4 NB of them, spilled)
        int ql = q;
        asm volatile("" : "+v"(ql));
        const double radius = sv[S_RADIUS];
        const double inv_radius = div_fast(1.0, radius);  // radius stays within [1e-32, 1e16]: no scaling cases
        // row q of Hs + diag(D^2), D^2 = clamp(diag, 1e-6, 1e32) / radius: the LM strategy (A.6)
        double arow[P], Lr[P], invd[P];
#pragma unroll
        for (int j = 0; j < P; ++j) arow[j] = Hs[qc * P + j];
        const double gsq = act ? gs[q] : 0.0;
        const double d2 = clampd(Hs[qc * P + qc], 1e-6, 1e32) * inv_radius;
        // Cholesky, one row per lane, column by column: s = a_ij - sum_k<j L_ik L_jk; lane j's s is the pivot
        bool ok = true;
#pragma unroll
        for (int j = 0; j < P; ++j) {
          double s_ = arow[j] + ((j == ql) ? d2 : 0.0);
#pragma unroll
          for (int kk = 0; kk < j; ++kk) s_ = fma(-Lr[kk], Lw[j * P + kk], s_);
          if (ql == j) bc[j] = s_;
          wave_lds_fence();
          const double dpiv = bc[j];
          ok = ok && (dpiv > 0.0) && isfinite(dpiv);
          const double inv = rsqrt_pos(fmax(dpiv, 1e-300));  // 1 / l_jj (never used when the pivot is not positive)
          invd[j] = inv;
          Lr[j] = s_ * inv;
          if (act && ql > j) Lw[q * P + j] = Lr[j];
          wave_lds_fence();
        }
        // forward substitution L y = gs, backward L^T z = y; the step is -z
        double accf = gsq, yq = 0.0;
#pragma unroll
        for (int kk = 0; kk < P; ++kk) {
          const double yk_own = accf * invd[kk];
          if (ql == kk) { bc[P + kk] = yk_own; yq = yk_own; }
          wave_lds_fence();
          const double yk = bc[P + kk];
          accf = fma((ql > kk) ? -Lr[kk] : 0.0, yk, accf);
        }
        double accb = yq, zq = 0.0;
#pragma unroll
        for (int kk = P - 1; kk >= 0; --kk) {
          const double zk_own = accb * invd[kk];
          if (ql == kk) { bc[2 * P + kk] = zk_own; zq = zk_own; }
          wave_lds_fence();
          const double zk = bc[2 * P + kk];
          accb = fma((act && ql < kk) ? -Lw[kk * P + qc] : 0.0, zk, accb);
        }
        const double stepq = act ? -zq : 0.0;
        bool valid = ok && !slot_any(act && !isfinite(stepq));
        double mcc = 0.0;
        if (valid) {
          if (act) bc[3 * P + q] = stepq;
          wave_lds_fence();
          double rowv = 0.0;
#pragma unroll
          for (int b = 0; b < P; ++b) rowv = fma(arow[b], bc[3 * P + b], rowv);
          double sg, sHs, u1;
          reduce3(4, stepq * gsq, act ? stepq * rowv : 0.0, 0.0, sg, sHs, u1);
          mcc = -sg - 0.5 * sHs;
          valid = mcc > 0.0;
        }
        if (!valid) {
          if (++R.num_invalid >= 5) { R.status = SMPC_FAILURE; R.reason = SMPC_REASON_INVALID_STEPS; R.phase = PH_DONE; break; }
          sv[S_RADIUS] = radius / sv[S_DECF]; sv[S_DECF] *= 2.0;
          wave_lds_fence();
          continue;
        }
        R.num_invalid = 0;
        sv[S_MCC] = mcc;
        double dq = 0.0, gq = 0.0;
        if (act) {
          dq = stepq * sc[q];
          dl[q] = dq;
          gq = gu[q] * dq;
          xt[q] = clampd(xc[q] + 1.0 * dq, lo_q, hi_q);
        }
        double gd0, u0, dirmax;
        reduce3(5, gq, 0.0, fabs(dq), gd0, u0, dirmax);
        sv[S_GD0] = gd0; sv[S_DIRMAX] = dirmax;
        sv[S_CUR_X] = 1.0;
        R.prev_vv = R.prev_gv = false;
        R.ls_iters = 0; R.n_samples = 1;
        R.phase = PH_LS;
        break;
      }
    }

    // ---------------------------------------------------------------- finished: outputs, a12 unpack
    if (R.phase == PH_DONE) {
      wave_lds_fence();
      const size_t s = c.scene;
      if (c.sl == 0) {
        if (k.o_status) k.o_status[s] = R.status;
        if (k.o_reason) k.o_reason[s] = R.reason;
        if (k.o_iterations) k.o_iterations[s] = R.iter;
        if (k.o_evaluations) k.o_evaluations[s] = R.evals;
        if (k.o_initial_cost) k.o_initial_cost[s] = sv[S_INITIAL_COST];
        if (k.o_final_cost) k.o_final_cost[s] = sv[S_COST];
      }
      if (k.o_params && c.sl < P) k.o_params[s * P + c.sl] = xc[c.sl];
      // saving_velocities[i], i = 0..T: block i/bl for i < CH, else the last block (src/optimizer.cpp:390-411); a scene
      // with a horizon of its own has Th + 1 entries, the rows behind them are written as zeros
      const int Th = hz.T;
      if (k.o_cmds) {
        for (int i = c.sl; i <= T; i += W) {
          const int b = block_of_step<NB>(i, hz);
          const bool in = i <= Th;
          k.o_cmds[(s * (T + 1) + i) * 2] = in ? xc[2 * b] : 0.0;
          k.o_cmds[(s * (T + 1) + i) * 2 + 1] = in ? xc[2 * b + 1] : 0.0;
        }
      }
      if (k.o_path) {
        // Re-roll (:420-446). The reference round-trips every heading through a quaternion (setRPY / getYaw), which
        // is the identity up to 1e-16 plus a wrap into (-pi, pi]; headings are produced here by the same sequential
        // adds followed by an exact wrap, then lane i integrates... positions need the sequential sums of
        // v cos(yaw_i) dt: lane i computes its own term, the running sum goes lane to lane in index order.
        const double* cst = c.lds + c.L.cst;
        double yaw = wrap_angle(cst[2]);
        double my_yaw_in = yaw, my_yaw_out = yaw;
        for (int i = 0; i <= T; ++i) {
          const int b = block_of_step<NB>(i, hz);
          const double nyaw = wrap_angle(yaw + xc[2 * b + 1] * k.dt);
          if (i == c.sl) { my_yaw_in = yaw; my_yaw_out = nyaw; }
          yaw = nyaw;
        }
        const int bi = block_of_step<NB>(c.sl, hz);
        double sn, cs;
        sincos(my_yaw_in, &sn, &cs);
        const double v = (c.sl <= T) ? xc[2 * bi] : 0.0;
        const double tx = v * cs * k.dt, ty = v * sn * k.dt;
        double px = cst[0], py = cst[1], mx = 0.0, my = 0.0;
        for (int i = 0; i <= T; ++i) {
          px += __shfl(tx, slot * W + i, 64);
          py += __shfl(ty, slot * W + i, 64);
          if (i == c.sl) { mx = px; my = py; }
        }
        if (c.sl <= T) {
          const bool in = c.sl <= Th;
          double* o = k.o_path + (s * (T + 1) + c.sl) * 3;
          o[0] = in ? mx : 0.0; o[1] = in ? my : 0.0; o[2] = in ? my_yaw_out : 0.0;
        }
      }
      R.phase = PH_FETCH;
    }
  }
#ifdef SMPC_STAMPS
  if (k.stamps && lane == 0) {
    for (int i = 0; i < 8; ++i) k.stamps[(size_t)blockIdx.x * 12 + i] = c.acc[i];
    for (int i = 0; i < 4; ++i) k.stamps[(size_t)blockIdx.x * 12 + 8 + i] = c.acc2[i];
  }
#endif
}

// K1 stand-alone: one sweep per scene at given parameters, rows written to HBM (parity checks, roofline runs).
// Up to three parameter blocks the sweep fits the 168 registers that three waves per SIMD allow (the headline shapes;
// K1 is a latency-bound streaming kernel, the third wave is worth 20 % of its time); beyond that the row buffers grow
// with P and the allocator is left alone.
#ifndef SMPC_EVAL_MIN_WAVES
#define SMPC_EVAL_MIN_WAVES(NB) ((NB) <= 3 ? 3 : 1)
#endif
template <int NB, int W, bool kVT = false>
__global__ __launch_bounds__(64, SMPC_EVAL_MIN_WAVES(NB)) void smpc_eval_kernel(const KParams) {
  const auto& k = *(KParamsK)__builtin_amdgcn_kernarg_segment_ptr();
  constexpr int P = 2 * NB;
  constexpr int S = kWave / W;
  extern __shared__ __attribute__((aligned(32))) double lds_all[];
  const int lane = threadIdx.x & 63;
  const int slot = lane / W;
  Ctx c;
  c.kp = &k;
  c.sl = lane - slot * W;
  c.L = make_layout(k.T, k.N, P, kLayoutEval, W);
  c.lds = lds_all + (size_t)slot * c.L.total;
  c.wave_lds = lds_all + (size_t)S * c.L.total;
  c.slot = slot;
  c.ag = k.people_rec;
  {
    double* atab = lds_all + atan_tab_offset(S * c.L.total, eval_extra_doubles(k.T, P, W));
    load_atan_nodes(c.kp, atab, lane);
    c.atab = atab;
  }
#ifdef SMPC_STAMPS
  for (int i = 0; i < 8; ++i) c.acc[i] = 0;
  for (int i = 0; i < 4; ++i) c.acc2[i] = 0;
  c.t_last = __builtin_amdgcn_s_memtime();
#endif
  const int scene_raw = blockIdx.x * S + slot;
  const bool live = scene_raw < k.B;
  const int scene = live ? scene_raw : k.B - 1;
  load_scene<W, kVT>(c, scene);
  wave_lds_fence();
  SMPC_STAMP(c, 0);
  const size_t s = scene;
  double* out_r = (live && k.e_residuals) ? k.e_residuals + s * k.e_M : nullptr;
  double* out_J = (live && k.e_jacobian) ? k.e_jacobian + s * (size_t)k.e_M * P : nullptr;
  const Horizon hz = get_horizon<NB, kVT>(c);
  if (!c.has_people || kVT) {
    // rows the scene does not have stay zero: the people rows of a scene without people, and — reference row order —
    // everything behind the 8 (or 5) Th + n_feasibility rows of a scene with a horizon of its own (in the critic-major
    // order the sweep itself writes the zero rows of every critic block it visits)
    const int Mb = (c.has_people ? 8 : 5) * (k.e_row_order == 1 ? k.T : hz.T) + (k.e_row_order == 1 ? k.nfeas : hz.nfeas);
    for (int i = Mb + c.sl; i < k.e_M; i += W) {
      if (out_r) out_r[i] = 0.0;
      if (out_J) for (int q = 0; q < P; ++q) out_J[(size_t)i * P + q] = 0.0;
    }
    if (kVT && k.e_row_order == 1) {  // critic-major: the feasibility rows the scene does not have
      const int base = (c.has_people ? 8 : 5) * k.T;
      for (int i = hz.nfeas + c.sl; i < k.nfeas; i += W) {
        if (out_r) out_r[base + i] = 0.0;
        if (out_J) for (int q = 0; q < P; ++q) out_J[(size_t)(base + i) * P + q] = 0.0;
      }
    }
  }
  const GramView G = sweep<NB, W, true, kVT>(c, k.e_x + s * P, out_r, out_J);
  if (live && c.sl == 0 && k.e_cost) k.e_cost[s] = 0.5 * G(P, P);
  if (live && k.e_gradient && c.sl < P) k.e_gradient[s * P + c.sl] = G(c.sl, P);
#ifdef SMPC_STAMPS
  SMPC_STAMP(c, 6);
  if (k.stamps && lane == 0) {
    for (int i = 0; i < 8; ++i) k.stamps[(size_t)blockIdx.x * 12 + i] = c.acc[i];
    for (int i = 0; i < 4; ++i) k.stamps[(size_t)blockIdx.x * 12 + 8 + i] = c.acc2[i];
  }
#endif
}

// Staging pass: people block of the reference layout ([T+1][6][N] per scene) -> the records the sweep reads
// ([N][T] x (px, py, vx, vy), written as whole 128-byte lines through LDS) + per-step valid mask and agent-angle tag.
// One slot per scene like the sweep kernels; once per people block (a solve re-reads the records ~50 times).
template <int W>
__global__ __launch_bounds__(64) void smpc_stage_kernel(const KParams) {
  SMPC_CHAIN_PRIORITY();
  const auto& k = *(KParamsK)__builtin_amdgcn_kernarg_segment_ptr();
  constexpr int S = kWave / W;
  extern __shared__ __attribute__((aligned(32))) double lds_all[];
  const int lane = threadIdx.x & 63;
  const int slot = lane / W, sl = lane - slot * W;
  const int T = k.T, N = k.N;
  const LdsLayout L = make_layout(T, N, 2, kLayoutStage, W);
  double* lds = lds_all + (size_t)slot * L.total;
  const int scene_raw = blockIdx.x * S + slot;
  const bool live = scene_raw < k.B;
  const int scene = live ? scene_raw : k.B - 1;
  const bool has_people = k.has_people ? k.has_people[scene] != 0 : true;
  double* ag = lds + L.ag;
  unsigned long long* vmask = reinterpret_cast<unsigned long long*>(lds + L.valid);
  double* aa = lds + L.lanec;
  if (has_people) stage_people<W>(&k, scene, sl, ag, vmask, aa);
  wave_lds_fence();
  if (live && has_people) {
    const size_t s = scene;
    const int nrec = N * T;
    v4d* dst = reinterpret_cast<v4d*>(k.stage_rec + s * (size_t)4 * nrec);
    const v4d* src = reinterpret_cast<const v4d*>(ag);
    for (int q = sl; q < nrec; q += W) dst[q] = src[q];  // consecutive lanes, consecutive 32-byte records
    if (sl < T) {
      double* aux = k.stage_aux + (s * T + sl) * 2;
      aux[0] = (lds + L.valid)[sl];
      aux[1] = aa[sl];
    }
  }
}

}  // namespace smpc
