// smpc_lm.hpp — per-wave Levenberg-Marquardt driver (the ceres::Solve call of reference src/optimizer.cpp:381,
// options :117-131) and the post-solve unpack (src/optimizer.cpp:390-446). Algorithm = Ceres' trust-region
// minimizer with bounds as specified in SURVEY.md Appendix A (A.4 .. A.11). Every quantity here is uniform
// across the 64 lanes of the wave; only sweep() is lane-parallel.
#pragma once

#include "smpc_device.hpp"

namespace smpc {

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
__device__ inline double interpolate_step(const Sample& lower, const Sample& previous, const Sample& current,
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

// In-register Cholesky solve of (Hs + diag(D2)) y = gs for P <= 20 (fully unrolled, packed lower triangle).
template <int P>
__device__ inline bool cholesky_solve(const double* Hs, const double* D2, const double* gs, double (&y)[P]) {
  double Lm[P * (P + 1) / 2];
  auto li = [](int i, int j) { return i * (i + 1) / 2 + j; };  // j <= i
  bool ok = true;
#pragma unroll
  for (int j = 0; j < P; ++j) {
    double d = Hs[j * P + j] + D2[j];
#pragma unroll
    for (int kk = 0; kk < j; ++kk) d -= Lm[li(j, kk)] * Lm[li(j, kk)];
    if (!(d > 0.0) || !isfinite(d)) ok = false;
    const double l = sqrt(d);
    Lm[li(j, j)] = l;
    const double inv = 1.0 / l;
#pragma unroll
    for (int i = j + 1; i < P; ++i) {
      double v = Hs[i * P + j];
#pragma unroll
      for (int kk = 0; kk < j; ++kk) v -= Lm[li(i, kk)] * Lm[li(j, kk)];
      Lm[li(i, j)] = v * inv;
    }
  }
#pragma unroll
  for (int i = 0; i < P; ++i) {
    double v = gs[i];
#pragma unroll
    for (int kk = 0; kk < i; ++kk) v -= Lm[li(i, kk)] * y[kk];
    y[i] = v / Lm[li(i, i)];
  }
#pragma unroll
  for (int i = P - 1; i >= 0; --i) {
    double v = y[i];
#pragma unroll
    for (int kk = i + 1; kk < P; ++kk) v -= Lm[li(kk, i)] * y[kk];
    y[i] = v / Lm[li(i, i)];
  }
  return ok;
}

__device__ inline double clampd(double v, double lo, double hi) { return fmin(fmax(v, lo), hi); }

// tf2 Quaternion::setRPY(0,0,yaw) followed by tf2::getYaw (x = y = 0): src/optimizer.cpp:434-439 round trips.
__device__ inline double yaw_roundtrip(double yaw) {
  double sz, cz;
  sincos(yaw * 0.5, &sz, &cz);
  return atan2(2.0 * (cz * sz), cz * cz - sz * sz);
}

template <int NB>
__device__ inline void solve_scene(Ctx<NB>& c) {
  constexpr int P = 2 * NB;
  const KParams& k = *c.kp;
  const smpc_params& prm = k.prm;
  const int lane = c.lane, T = k.T;
  double* lm = c.lds + c.L.lm;
  double* Hs = lm;                 // [P*P] scaled J^T J at the current point
  double* gs = Hs + P * P;         // [P]   scaled gradient
  double* gu = gs + P;             // [P]   unscaled gradient
  double* scratch = c.lds + c.L.scratch;

  double lo[P], hi[P], x[P], scale[P];
#pragma unroll
  for (int q = 0; q < P; ++q) { lo[q] = -1.7976931348623157e308; hi[q] = 1.7976931348623157e308; }
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    if (b < k.nbounded) { lo[2 * b] = prm.v_min; hi[2 * b] = prm.v_max; lo[2 * b + 1] = prm.w_min; hi[2 * b + 1] = prm.w_max; }
  }
  const double* xin = k.init_params + (size_t)c.scene * P;
#pragma unroll
  for (int q = 0; q < P; ++q) x[q] = clampd(xin[q] + 0.0, lo[q], hi[q]);  // Plus(x, 0): project the start point
  double x_norm = 0.0;
#pragma unroll
  for (int q = 0; q < P; ++q) x_norm += x[q] * x[q];
  x_norm = sqrt(x_norm);

  int status = SMPC_NO_CONVERGENCE, reason = SMPC_REASON_MAX_ITERATIONS, iter = 0, evals = 0;
  double cost = 0.0, initial_cost = 0.0, gmax = 0.0;
  {
    Gram<P> G;
    bool finite;
    sweep<NB>(c, x, G, finite, nullptr, nullptr);
    ++evals;
    cost = 0.5 * G.v[Gram<P>::idx(P, P)];
    initial_cost = cost;
    if (!finite) { status = SMPC_FAILURE; reason = SMPC_REASON_EVAL_FAILED; }
#pragma unroll
    for (int q = 0; q < P; ++q) scale[q] = 1.0 / (1.0 + sqrt(G.v[Gram<P>::idx(q, q)]));  // Jacobi scaling, fixed at iteration 0
#pragma unroll
    for (int a = 0; a < P; ++a) {
#pragma unroll
      for (int b = 0; b < P; ++b) Hs[a * P + b] = G.H(a, b) * scale[a] * scale[b];
      const double g = G.v[Gram<P>::idx(a, P)];
      gu[a] = g; gs[a] = g * scale[a];
      gmax = fmax(gmax, fabs(x[a] - clampd(x[a] - g, lo[a], hi[a])));
    }
    __syncthreads();
  }

  if (status != SMPC_FAILURE) {
    double radius = 1e4, decrease_factor = 2.0;
    int num_invalid = 0;
    bool step_successful = true, at_least_one = false;
    for (;;) {
      if (iter >= prm.max_iterations) { status = SMPC_NO_CONVERGENCE; reason = SMPC_REASON_MAX_ITERATIONS; break; }
      if (step_successful && gmax <= prm.gradient_tol && !prm.fixed_iterations) { status = SMPC_CONVERGENCE; reason = SMPC_REASON_GRADIENT_TOL; break; }
      if (radius <= 1e-32) { status = SMPC_CONVERGENCE; reason = SMPC_REASON_MIN_RADIUS; break; }
      ++iter;
      step_successful = false;

      // LM step: (Hs + D^2) y = gs, step = -y (A.6); model cost change (A.7)
      double step[P], D2[P];
#pragma unroll
      for (int q = 0; q < P; ++q) {
        const double d = sqrt(clampd(Hs[q * P + q], 1e-6, 1e32) / radius);
        D2[q] = d * d;
      }
      bool valid = cholesky_solve<P>(Hs, D2, gs, step);
      double mcc = 0.0;
#pragma unroll
      for (int q = 0; q < P; ++q) { if (!isfinite(step[q])) valid = false; step[q] = -step[q]; }
      if (valid) {
        double sg = 0.0, sHs = 0.0;
#pragma unroll
        for (int a = 0; a < P; ++a) {
          sg += step[a] * gs[a];
          double row = 0.0;
#pragma unroll
          for (int b = 0; b < P; ++b) row += Hs[a * P + b] * step[b];
          sHs += step[a] * row;
        }
        mcc = -sg - 0.5 * sHs;
        valid = mcc > 0.0;
      }
      if (!valid) {
        if (++num_invalid >= 5) { status = SMPC_FAILURE; reason = SMPC_REASON_INVALID_STEPS; break; }
        radius = radius / decrease_factor; decrease_factor *= 2.0;
        continue;
      }
      num_invalid = 0;
      double delta[P];
#pragma unroll
      for (int q = 0; q < P; ++q) delta[q] = step[q] * scale[q];

      // projected Armijo line search (A.8); every sample is a full sweep (value + gradient, CUBIC interpolation)
      double gd0 = 0.0, dirmax = 0.0;
#pragma unroll
      for (int q = 0; q < P; ++q) { gd0 += gu[q] * delta[q]; dirmax = fmax(dirmax, fabs(delta[q])); }
      Sample lower{0.0, cost, gd0, true, true};
      Sample previous{0.0, 0.0, 0.0, false, false}, current{1.0, 0.0, 0.0, false, false};
      double xt[P];
      Gram<P> G;
      bool finite;
      auto sample_at = [&](double alpha, Sample& s) {
#pragma unroll
        for (int q = 0; q < P; ++q) xt[q] = clampd(x[q] + alpha * delta[q], lo[q], hi[q]);
        sweep<NB>(c, xt, G, finite, nullptr, nullptr);
        ++evals;
        s.x = alpha;
        s.value = 0.5 * G.v[Gram<P>::idx(P, P)];
        s.value_valid = finite && isfinite(s.value);
        double gd = 0.0;
#pragma unroll
        for (int q = 0; q < P; ++q) gd += delta[q] * G.v[Gram<P>::idx(q, P)];
        s.gradient = gd;
        s.gradient_valid = s.value_valid && isfinite(gd);
      };
      sample_at(1.0, current);
      bool ls_ok = false;
      int ls_iters = 0, n_samples = 1;
      for (;;) {
        if (current.value_valid && !(current.value > cost + 1e-4 * gd0 * current.x)) { ls_ok = true; break; }
        ++ls_iters;
        if (ls_iters >= 20) break;
        const double step_size = interpolate_step(lower, previous, current, 1e-3 * current.x, 0.6 * current.x, scratch);
        if (step_size * dirmax < 1e-9) break;
        previous = current;
        sample_at(step_size, current);
        ++n_samples;
      }
      if (ls_ok) {
#pragma unroll
        for (int q = 0; q < P; ++q) delta[q] *= current.x;
      } else if (n_samples > 1) {
        sample_at(1.0, current);  // line search failed: the candidate is the full step again
      }
      const double cand_cost = current.value_valid ? current.value : 1.7976931348623157e308;
      double step_norm = 0.0;
#pragma unroll
      for (int q = 0; q < P; ++q) step_norm += (x[q] - xt[q]) * (x[q] - xt[q]);
      step_norm = sqrt(step_norm);
      const bool tol_allowed = !prm.fixed_iterations && (!prm.tol_needs_successful_step || at_least_one);
      if (tol_allowed && step_norm <= prm.param_tol * (x_norm + prm.param_tol)) { status = SMPC_CONVERGENCE; reason = SMPC_REASON_PARAMETER_TOL; break; }
      const double cost_change = cost - cand_cost;
      if (tol_allowed && fabs(cost_change) <= prm.fn_tol * cost) { status = SMPC_CONVERGENCE; reason = SMPC_REASON_FUNCTION_TOL; break; }
      const double rho = (cand_cost >= 1.7976931348623157e308) ? -1.7976931348623157e308 : cost_change / mcc;
      if (rho > 1e-3) {
        x_norm = 0.0;
        gmax = 0.0;
        __syncthreads();
#pragma unroll
        for (int a = 0; a < P; ++a) {
          x[a] = xt[a];
          x_norm += x[a] * x[a];
#pragma unroll
          for (int b = 0; b < P; ++b) Hs[a * P + b] = G.H(a, b) * scale[a] * scale[b];
          const double g = G.v[Gram<P>::idx(a, P)];
          gu[a] = g; gs[a] = g * scale[a];
          gmax = fmax(gmax, fabs(x[a] - clampd(x[a] - g, lo[a], hi[a])));
        }
        __syncthreads();
        x_norm = sqrt(x_norm);
        cost = cand_cost;
        step_successful = true; at_least_one = true;
        const double t = 2.0 * rho - 1.0;
        radius = radius / fmax(1.0 / 3.0, 1.0 - t * t * t);
        radius = fmin(1e16, radius);
        decrease_factor = 2.0;
      } else {
        radius = radius / decrease_factor; decrease_factor *= 2.0;
      }
    }
  }

  // ---- outputs: params, a12 unpack (src/optimizer.cpp:390-446)
  const size_t s = c.scene;
  if (lane == 0) {
    if (k.o_status) k.o_status[s] = status;
    if (k.o_reason) k.o_reason[s] = reason;
    if (k.o_iterations) k.o_iterations[s] = iter;
    if (k.o_evaluations) k.o_evaluations[s] = evals;
    if (k.o_initial_cost) k.o_initial_cost[s] = initial_cost;
    if (k.o_final_cost) k.o_final_cost[s] = cost;
  }
  if (k.o_params && lane < P) {
    double v = 0.0;
#pragma unroll
    for (int q = 0; q < P; ++q) v = (q == lane) ? x[q] : v;
    k.o_params[s * P + lane] = v;
  }
  // saving_velocities[i], i = 0..T: block i/bl for i < CH, else the last block (:390-411)
  const int blast = (k.CH - 1) / k.bl;
  if (k.o_cmds) {
    for (int i = lane; i <= T; i += kWave) {
      const int b = (i < k.CH) ? i / k.bl : blast;
      double v = 0.0, w = 0.0;
#pragma unroll
      for (int q = 0; q < NB; ++q) { v = (q == b) ? x[2 * q] : v; w = (q == b) ? x[2 * q + 1] : w; }
      k.o_cmds[(s * (T + 1) + i) * 2] = v;
      k.o_cmds[(s * (T + 1) + i) * 2 + 1] = w;
    }
  }
  if (k.o_path) {
    // sequential re-roll with the reference's quaternion round trips; uniform, lane 0 stores
    double px = c.x0, py = c.y0, yaw = yaw_roundtrip(c.yaw0);
    for (int i = 0; i <= T; ++i) {
      const int b = (i < k.CH) ? i / k.bl : blast;
      double v = 0.0, w = 0.0;
#pragma unroll
      for (int q = 0; q < NB; ++q) { v = (q == b) ? x[2 * q] : v; w = (q == b) ? x[2 * q + 1] : w; }
      double sn, cs;
      sincos(yaw, &sn, &cs);
      px = px + v * cs * k.dt;
      py = py + v * sn * k.dt;
      yaw = yaw_roundtrip(yaw + w * k.dt);
      if (lane == 0) {
        double* o = k.o_path + (s * (T + 1) + i) * 3;
        o[0] = px; o[1] = py; o[2] = yaw;
      }
    }
  }
}

}  // namespace smpc
