// smpc_oracle.cpp — CPU restatement of the reference's MPC inner loop.
//
// *** TEST INFRASTRUCTURE ONLY. ***  Nothing under nav2_social_mpc_controller_amd/ may include, link or call
// this file. Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, and only as the
// checker / reported CPU baseline.
//
// *** PARITY UNPINNED. ***  The reference ships no tests, fixtures or golden vectors, and
// its arithmetic core is Ceres Solver, which is absent from /root/reference and from this image (un-vendored,
// un-pinned: package.xml:27, CMakeLists.txt:21). This file therefore restates
//   (1) the reference's own cost functors, line by line cited below, on a forward-mode dual number that follows
//       the published semantics of ceres::Jet<double,4> / DynamicAutoDiffCostFunction (stride 4), and
//   (2) the published algorithm of Ceres' trust-region Levenberg-Marquardt minimizer (Ceres 2.0.x:
//       docs/nnls_solving + internal/ceres/{trust_region_minimizer,levenberg_marquardt_strategy,line_search,
//       polynomial,parameter_block}.cc semantics), anchored on the reference's call sites
//       src/optimizer.cpp:117-131 (options) and :241-381 (problem).
// It is cross-checked by an independent Python/torch.float64 restatement (oracle/pyref, tests/golden).
//
// Reference citations are relative to /root/reference.

#include <algorithm>
#include <cmath>
#include <complex>
#include <cstdint>
#include <cstring>
#include <limits>
#include <thread>
#include <vector>

#include "../include/smpc.h"

namespace {

// ------------------------------------------------------------------------------------------------
// Forward-mode dual number: value + 4 tangents. Semantics follow ceres::Jet<double,4> (published
// ceres/jet.h behaviour: comparisons act on the scalar part; f/g is computed as f * (1/g)).
// ------------------------------------------------------------------------------------------------
constexpr int kStride = 4;  // DynamicAutoDiffCostFunction<F, Stride = 4>

// Operation counting (the FP64 work of one Jacobian evaluation in the reference's formulation, i.e. what
// DynamicAutoDiffCostFunction executes on Jets). Compiled in only with -DSMPC_ORACLE_COUNT_OPS (a second library,
// oracle/_build/libsmpc_oracle_count.so); the timed oracle carries none of it. Counts are per scalar operation:
// a Jet product is 1 + 2*4 multiplies and 4 adds, and so on.
#ifdef SMPC_ORACLE_COUNT_OPS
struct OpCount { long add, mul, div, sqrt, exp, sincos, atan2; };
thread_local OpCount g_ops = {0, 0, 0, 0, 0, 0, 0};
#define SMPC_OPS(field, n) (g_ops.field += (n))
#else
#define SMPC_OPS(field, n) ((void)0)
#endif

struct Jet {
  double a;
  double v[kStride];
  Jet() : a(0.0) { for (double& t : v) t = 0.0; }
  Jet(double s) : a(s) { for (double& t : v) t = 0.0; }  // NOLINT implicit, like ceres::Jet
};

inline Jet operator+(const Jet& f, const Jet& g) { SMPC_OPS(add, 1 + kStride); Jet h; h.a = f.a + g.a; for (int k = 0; k < kStride; ++k) h.v[k] = f.v[k] + g.v[k]; return h; }
inline Jet operator-(const Jet& f, const Jet& g) { SMPC_OPS(add, 1 + kStride); Jet h; h.a = f.a - g.a; for (int k = 0; k < kStride; ++k) h.v[k] = f.v[k] - g.v[k]; return h; }
inline Jet operator-(const Jet& f) { Jet h; h.a = -f.a; for (int k = 0; k < kStride; ++k) h.v[k] = -f.v[k]; return h; }
inline Jet operator*(const Jet& f, const Jet& g) { SMPC_OPS(mul, 1 + 2 * kStride); SMPC_OPS(add, kStride); Jet h; h.a = f.a * g.a; for (int k = 0; k < kStride; ++k) h.v[k] = f.a * g.v[k] + f.v[k] * g.a; return h; }
inline Jet operator/(const Jet& f, const Jet& g) {
  SMPC_OPS(div, 1); SMPC_OPS(mul, 1 + 2 * kStride); SMPC_OPS(add, kStride);
  const double g_inv = 1.0 / g.a;
  const double q = f.a * g_inv;
  Jet h; h.a = q;
  for (int k = 0; k < kStride; ++k) h.v[k] = (f.v[k] - q * g.v[k]) * g_inv;
  return h;
}
inline Jet operator+(const Jet& f, double s) { SMPC_OPS(add, 1); Jet h = f; h.a = f.a + s; return h; }
inline Jet operator+(double s, const Jet& f) { SMPC_OPS(add, 1); Jet h = f; h.a = f.a + s; return h; }
inline Jet operator-(const Jet& f, double s) { SMPC_OPS(add, 1); Jet h = f; h.a = f.a - s; return h; }
inline Jet operator-(double s, const Jet& f) { SMPC_OPS(add, 1); Jet h; h.a = s - f.a; for (int k = 0; k < kStride; ++k) h.v[k] = -f.v[k]; return h; }
inline Jet operator*(const Jet& f, double s) { SMPC_OPS(mul, 1 + kStride); Jet h; h.a = f.a * s; for (int k = 0; k < kStride; ++k) h.v[k] = f.v[k] * s; return h; }
inline Jet operator*(double s, const Jet& f) { return f * s; }
inline Jet operator/(const Jet& f, double s) { SMPC_OPS(div, 1); SMPC_OPS(mul, 1 + kStride); const double si = 1.0 / s; Jet h; h.a = f.a * si; for (int k = 0; k < kStride; ++k) h.v[k] = f.v[k] * si; return h; }
inline Jet operator/(double s, const Jet& g) { SMPC_OPS(div, 2); SMPC_OPS(mul, 1 + kStride); const double m = -s / (g.a * g.a); Jet h; h.a = s / g.a; for (int k = 0; k < kStride; ++k) h.v[k] = g.v[k] * m; return h; }
inline Jet& operator+=(Jet& f, const Jet& g) { f = f + g; return f; }
inline Jet& operator-=(Jet& f, const Jet& g) { f = f - g; return f; }
inline Jet& operator+=(Jet& f, double s) { SMPC_OPS(add, 1); f.a += s; return f; }
inline Jet& operator-=(Jet& f, double s) { SMPC_OPS(add, 1); f.a -= s; return f; }
inline bool operator<(const Jet& f, const Jet& g) { return f.a < g.a; }
inline bool operator>(const Jet& f, const Jet& g) { return f.a > g.a; }
inline bool operator<=(const Jet& f, const Jet& g) { return f.a <= g.a; }
inline bool operator>=(const Jet& f, const Jet& g) { return f.a >= g.a; }
inline bool operator==(const Jet& f, const Jet& g) { return f.a == g.a; }
inline bool operator<(const Jet& f, double s) { return f.a < s; }
inline bool operator>(const Jet& f, double s) { return f.a > s; }
inline bool operator<=(const Jet& f, double s) { return f.a <= s; }
inline bool operator>=(const Jet& f, double s) { return f.a >= s; }
inline bool operator==(const Jet& f, double s) { return f.a == s; }

inline double Sqrt(double x) { return std::sqrt(x); }
inline double Exp(double x) { return std::exp(x); }
inline double Sin(double x) { return std::sin(x); }
inline double Cos(double x) { return std::cos(x); }
inline double Atan2(double y, double x) { return std::atan2(y, x); }
inline Jet Sqrt(const Jet& f) { SMPC_OPS(sqrt, 1); SMPC_OPS(div, 1); SMPC_OPS(mul, 1 + kStride); const double t = std::sqrt(f.a); const double m = 1.0 / (2.0 * t); Jet h; h.a = t; for (int k = 0; k < kStride; ++k) h.v[k] = f.v[k] * m; return h; }
inline Jet Exp(const Jet& f) { SMPC_OPS(exp, 1); SMPC_OPS(mul, kStride); const double t = std::exp(f.a); Jet h; h.a = t; for (int k = 0; k < kStride; ++k) h.v[k] = t * f.v[k]; return h; }
inline Jet Sin(const Jet& f) { SMPC_OPS(sincos, 2); SMPC_OPS(mul, kStride); const double c = std::cos(f.a); Jet h; h.a = std::sin(f.a); for (int k = 0; k < kStride; ++k) h.v[k] = c * f.v[k]; return h; }
inline Jet Cos(const Jet& f) { SMPC_OPS(sincos, 2); SMPC_OPS(mul, kStride); const double s = -std::sin(f.a); Jet h; h.a = std::cos(f.a); for (int k = 0; k < kStride; ++k) h.v[k] = s * f.v[k]; return h; }
inline Jet Atan2(const Jet& g, const Jet& f) {  // atan2(y = g, x = f)
  SMPC_OPS(atan2, 1); SMPC_OPS(div, 1); SMPC_OPS(mul, 2 + 3 * kStride); SMPC_OPS(add, 1 + kStride);
  const double t = 1.0 / (f.a * f.a + g.a * g.a);
  Jet h; h.a = std::atan2(g.a, f.a);
  for (int k = 0; k < kStride; ++k) h.v[k] = t * (-g.a * f.v[k] + f.a * g.v[k]);
  return h;
}
inline double Value(double x) { return x; }
inline double Value(const Jet& x) { return x.a; }
template <typename T> inline T MaxOf();
template <> inline double MaxOf<double>() { return std::numeric_limits<double>::max(); }
template <> inline Jet MaxOf<Jet>() { return Jet(std::numeric_limits<double>::max()); }

#include "smpc_functors.inc"  // the functors, templated on the scalar type (shared with oracle/ceres_harness.cpp)

// the dual-number overloads the functors find by argument-dependent lookup
inline void ZeroValue(Jet& x) { x.a = 0.0; }
inline Jet InterpEval(const Scene& s, const Jet& r, const Jet& c) {
  double f, dfdr, dfdc;
  BiCubic(s, r.a, c.a, &f, &dfdr, &dfdc);
  Jet h; h.a = f;
  for (int k = 0; k < kStride; ++k) h.v[k] = dfdr * r.v[k] + dfdc * c.v[k];
  return h;
}

// Evaluate all residual blocks at x. jac (M x P, row-major, may be null) is filled the way
// DynamicAutoDiffCostFunction does it: ceil(n_visible_params / 4) dual passes, residual value from the dual
// pass when Jacobians are requested, from the plain double pass otherwise.
bool Evaluate(const Scene& s, const std::vector<Block>& blocks, const double* x, double* residuals, double* jac,
              double* cost_out, double* gradient) {
  const Dims& d = s.d;
  const int P = d.P;
  double cost = 0.0;
  if (gradient) std::fill(gradient, gradient + P, 0.0);
  bool ok = true;
  std::vector<double> row(P);
  for (size_t k = 0; k < blocks.size(); ++k) {
    const Block& b = blocks[k];
    double r = 0.0;
    std::fill(row.begin(), row.end(), 0.0);
    if (b.kind == kVelFeas) {
      // AutoDiffCostFunction<VelocityFeasibilityCost, 1, 2, 2>: state1 = block i, state2 = block i-1 (:368-369)
      if (!jac && !gradient) {
        r = VelocityFeasibilityResidual<double>(s, x + 2 * b.i, x + 2 * (b.i - 1), b.i);
      } else {
        Jet s1[2], s2[2];
        s1[0] = Jet(x[2 * b.i]); s1[0].v[0] = 1.0;
        s1[1] = Jet(x[2 * b.i + 1]); s1[1].v[1] = 1.0;
        s2[0] = Jet(x[2 * (b.i - 1)]); s2[0].v[2] = 1.0;
        s2[1] = Jet(x[2 * (b.i - 1) + 1]); s2[1].v[3] = 1.0;
        Jet rj = VelocityFeasibilityResidual<Jet>(s, s1, s2, b.i);
        r = rj.a;
        row[2 * b.i] = rj.v[0]; row[2 * b.i + 1] = rj.v[1];
        row[2 * (b.i - 1)] = rj.v[2]; row[2 * (b.i - 1) + 1] = rj.v[3];
      }
    } else {
      // visible parameter blocks: 0..(i<CH ? i/bl : (CH-1)/bl)   (src/optimizer.cpp:271-288 etc.)
      const int nvis_blocks = ((b.i < d.CH) ? b.i / d.bl : (d.CH - 1) / d.bl) + 1;
      const int nvis = 2 * nvis_blocks;
      if (!jac && !gradient) {
        const double* pp[SMPC_MAX_BLOCKS];
        for (int q = 0; q < nvis_blocks; ++q) pp[q] = x + 2 * q;
        r = EvalDynamic<double>(s, b, pp);
      } else {
        Jet xs[2 * SMPC_MAX_BLOCKS];
        const Jet* pp[SMPC_MAX_BLOCKS];
        for (int q = 0; q < nvis_blocks; ++q) pp[q] = xs + 2 * q;
        for (int start = 0; start < nvis; start += kStride) {
          for (int q = 0; q < nvis; ++q) {
            xs[q] = Jet(x[q]);
            if (q >= start && q < start + kStride) xs[q].v[q - start] = 1.0;
          }
          Jet rj = EvalDynamic<Jet>(s, b, pp);
          r = rj.a;
          for (int q = start; q < std::min(start + kStride, nvis); ++q) row[q] = rj.v[q - start];
        }
      }
    }
    if (!std::isfinite(r)) ok = false;
    for (int q = 0; q < P; ++q) if (!std::isfinite(row[q])) ok = false;
    if (residuals) residuals[k] = r;
    if (jac) std::memcpy(jac + k * P, row.data(), sizeof(double) * P);
    cost += 0.5 * r * r;
    if (gradient) for (int q = 0; q < P; ++q) gradient[q] += row[q] * r;
  }
  *cost_out = cost;
  return ok;
}

// ------------------------------------------------------------------------------------------------
// Small dense linear algebra used by the step solvers.
// ------------------------------------------------------------------------------------------------
// In-place Cholesky A = L L^T (lower), n x n row-major. Returns false if not positive definite.
bool Cholesky(std::vector<double>& A, int n) {
  for (int j = 0; j < n; ++j) {
    double djj = A[j * n + j];
    for (int k = 0; k < j; ++k) djj -= A[j * n + k] * A[j * n + k];
    if (!(djj > 0.0) || !std::isfinite(djj)) return false;
    const double ljj = std::sqrt(djj);
    A[j * n + j] = ljj;
    for (int i = j + 1; i < n; ++i) {
      double v = A[i * n + j];
      for (int k = 0; k < j; ++k) v -= A[i * n + k] * A[j * n + k];
      A[i * n + j] = v / ljj;
    }
  }
  return true;
}
void CholeskySolve(const std::vector<double>& L, int n, double* b) {
  for (int i = 0; i < n; ++i) { double v = b[i]; for (int k = 0; k < i; ++k) v -= L[i * n + k] * b[k]; b[i] = v / L[i * n + i]; }
  for (int i = n - 1; i >= 0; --i) { double v = b[i]; for (int k = i + 1; k < n; ++k) v -= L[k * n + i] * b[k]; b[i] = v / L[i * n + i]; }
}

// Least squares min ||A y - b|| via Householder QR; A is m x n row-major (destroyed), b length m (destroyed).
bool HouseholderSolve(std::vector<double>& A, std::vector<double>& b, int m, int n, double* y) {
  for (int k = 0; k < n; ++k) {
    double norm = 0.0;
    for (int i = k; i < m; ++i) norm += A[i * n + k] * A[i * n + k];
    norm = std::sqrt(norm);
    if (norm == 0.0) return false;
    const double alpha = (A[k * n + k] > 0.0) ? -norm : norm;
    std::vector<double> v(m - k);
    for (int i = k; i < m; ++i) v[i - k] = A[i * n + k];
    v[0] -= alpha;
    double vnorm2 = 0.0;
    for (double t : v) vnorm2 += t * t;
    if (vnorm2 == 0.0) continue;
    for (int j = k; j < n; ++j) {
      double dot = 0.0;
      for (int i = k; i < m; ++i) dot += v[i - k] * A[i * n + j];
      const double f = 2.0 * dot / vnorm2;
      for (int i = k; i < m; ++i) A[i * n + j] -= f * v[i - k];
    }
    double dot = 0.0;
    for (int i = k; i < m; ++i) dot += v[i - k] * b[i];
    const double f = 2.0 * dot / vnorm2;
    for (int i = k; i < m; ++i) b[i] -= f * v[i - k];
  }
  for (int i = n - 1; i >= 0; --i) {
    double v = b[i];
    for (int k = i + 1; k < n; ++k) v -= A[i * n + k] * y[k];
    y[i] = v / A[i * n + i];
  }
  return true;
}

// General dense solve with full pivoting (FindInterpolatingPolynomial uses a full-pivot LU). n <= 6.
bool FullPivSolve(std::vector<double> A, std::vector<double> b, int n, double* x) {
  std::vector<int> colperm(n);
  for (int i = 0; i < n; ++i) colperm[i] = i;
  for (int k = 0; k < n; ++k) {
    int pr = k, pc = k; double best = -1.0;
    for (int i = k; i < n; ++i) for (int j = k; j < n; ++j) { const double v = std::fabs(A[i * n + j]); if (v > best) { best = v; pr = i; pc = j; } }
    if (best == 0.0) { for (int i = k; i < n; ++i) b[i] = 0.0; break; }
    if (pr != k) { for (int j = 0; j < n; ++j) std::swap(A[pr * n + j], A[k * n + j]); std::swap(b[pr], b[k]); }
    if (pc != k) { for (int i = 0; i < n; ++i) std::swap(A[i * n + pc], A[i * n + k]); std::swap(colperm[pc], colperm[k]); }
    for (int i = k + 1; i < n; ++i) {
      const double f = A[i * n + k] / A[k * n + k];
      for (int j = k; j < n; ++j) A[i * n + j] -= f * A[k * n + j];
      b[i] -= f * b[k];
    }
  }
  std::vector<double> z(n, 0.0);
  for (int i = n - 1; i >= 0; --i) {
    if (A[i * n + i] == 0.0) { z[i] = 0.0; continue; }
    double v = b[i];
    for (int k = i + 1; k < n; ++k) v -= A[i * n + k] * z[k];
    z[i] = v / A[i * n + i];
  }
  for (int i = 0; i < n; ++i) x[colperm[i]] = z[i];
  return true;
}

// ------------------------------------------------------------------------------------------------
// Polynomials (coefficients highest degree first), as used by the Armijo line search's interpolation.
// ------------------------------------------------------------------------------------------------
long g_aberth_calls = 0, g_aberth_iters = 0;  // diagnostics (single-threaded runs only)
using Poly = std::vector<double>;
inline double EvalPoly(const Poly& p, double x) { double v = 0.0; for (double c : p) v = v * x + c; return v; }
Poly DiffPoly(const Poly& p) {
  const int deg = static_cast<int>(p.size()) - 1;
  if (deg == 0) return Poly{0.0};
  Poly d(deg);
  for (int i = 0; i < deg; ++i) d[i] = (deg - i) * p[i];
  return d;
}
// Real parts of all roots (complex ones included, like the reference implementation's "overkill" remark).
bool PolyRootsReal(Poly p, std::vector<double>* real) {
  real->clear();
  size_t lead = 0;
  while (lead + 1 < p.size() && p[lead] == 0.0) ++lead;
  p.erase(p.begin(), p.begin() + lead);
  const int deg = static_cast<int>(p.size()) - 1;
  if (p.empty()) return false;
  if (deg == 0) return true;
  if (deg == 1) { real->push_back(-p[1] / p[0]); return true; }
  if (deg == 2) {
    const double a = p[0], b = p[1], c = p[2];
    const double D = b * b - 4 * a * c;
    const double sqrt_D = std::sqrt(std::fabs(D));
    if (D >= 0) {
      if (b >= 0) { real->push_back((-b - sqrt_D) / (2.0 * a)); real->push_back((2.0 * c) / (-b - sqrt_D)); }
      else { real->push_back((2.0 * c) / (-b + sqrt_D)); real->push_back((-b + sqrt_D) / (2.0 * a)); }
    } else { real->push_back(-b / (2.0 * a)); real->push_back(-b / (2.0 * a)); }
    return true;
  }
  // deg >= 3: Aberth-Ehrlich simultaneous iteration on the monic polynomial (stands in for the
  // companion-matrix eigenvalue solve; both return all complex roots to round-off).
  using C = std::complex<double>;
  std::vector<C> c(deg + 1);
  for (int i = 0; i <= deg; ++i) c[i] = p[i] / p[0];
  double radius = 0.0;
  for (int i = 1; i <= deg; ++i) radius = std::max(radius, std::pow(std::fabs(p[i] / p[0]), 1.0 / i));
  radius = std::max(2.0 * radius, 1e-300);
  std::vector<C> z(deg);
  for (int i = 0; i < deg; ++i) z[i] = std::polar(radius, 2.0 * M_PI * i / deg + 0.4);
  ++g_aberth_calls;
  for (int it = 0; it < 200; ++it) {
    ++g_aberth_iters;
    double maxstep = 0.0;
    for (int i = 0; i < deg; ++i) {
      C pv = c[0], dv = 0.0;
      for (int k = 1; k <= deg; ++k) { dv = dv * z[i] + pv; pv = pv * z[i] + c[k]; }
      if (pv == C(0.0)) continue;
      C ratio = pv / dv;
      C sum = 0.0;
      for (int j = 0; j < deg; ++j) if (j != i) sum += 1.0 / (z[i] - z[j]);
      C step = ratio / (1.0 - ratio * sum);
      z[i] -= step;
      maxstep = std::max(maxstep, std::abs(step) / std::max(1e-300, std::abs(z[i])));
    }
    if (maxstep < 1e-15) break;
  }
  for (int i = 0; i < deg; ++i) real->push_back(z[i].real());
  return true;
}

struct FunctionSample {
  double x = 0.0, value = 0.0, gradient = 0.0;
  bool value_is_valid = false, gradient_is_valid = false;
};

Poly FindInterpolatingPolynomial(const std::vector<FunctionSample>& samples) {
  int nc = 0;
  for (const auto& s : samples) { if (s.value_is_valid) ++nc; if (s.gradient_is_valid) ++nc; }
  const int degree = nc - 1;
  std::vector<double> lhs(nc * nc, 0.0), rhs(nc, 0.0);
  int row = 0;
  for (const auto& s : samples) {
    if (s.value_is_valid) {
      for (int j = 0; j <= degree; ++j) lhs[row * nc + j] = std::pow(s.x, degree - j);
      rhs[row] = s.value; ++row;
    }
    if (s.gradient_is_valid) {
      for (int j = 0; j < degree; ++j) lhs[row * nc + j] = (degree - j) * std::pow(s.x, degree - j - 1);
      rhs[row] = s.gradient; ++row;
    }
  }
  Poly p(nc, 0.0);
  FullPivSolve(lhs, rhs, nc, p.data());
  return p;
}

void MinimizePolynomial(const Poly& poly, double x_min, double x_max, double* optimal_x, double* optimal_value) {
  *optimal_x = (x_min + x_max) / 2.0;
  *optimal_value = EvalPoly(poly, *optimal_x);
  const double vmin = EvalPoly(poly, x_min);
  if (vmin < *optimal_value) { *optimal_value = vmin; *optimal_x = x_min; }
  const double vmax = EvalPoly(poly, x_max);
  if (vmax < *optimal_value) { *optimal_value = vmax; *optimal_x = x_max; }
  if (poly.size() <= 2) return;
  std::vector<double> roots;
  if (!PolyRootsReal(DiffPoly(poly), &roots)) return;
  for (double root : roots) {
    if (root < x_min || root > x_max) continue;
    const double v = EvalPoly(poly, root);
    if (v < *optimal_value) { *optimal_value = v; *optimal_x = root; }
  }
}

void MinimizeInterpolatingPolynomial(const std::vector<FunctionSample>& samples, double x_min, double x_max,
                                     double* optimal_x, double* optimal_value) {
  const Poly poly = FindInterpolatingPolynomial(samples);
  MinimizePolynomial(poly, x_min, x_max, optimal_x, optimal_value);
  for (const auto& s : samples) {
    if (s.x < x_min || s.x > x_max) continue;
    const double v = EvalPoly(poly, s.x);
    if (v < *optimal_value) { *optimal_x = s.x; *optimal_value = v; }
  }
}

// ------------------------------------------------------------------------------------------------
// a11  ceres::Solve — trust-region Levenberg-Marquardt with bounds.
// ------------------------------------------------------------------------------------------------
struct TraceRow { double iter, cost, cost_change, gradient_max_norm, step_norm, rho, radius, ls_evals, accepted; };

struct SolveResult {
  int status = SMPC_FAILURE, reason = SMPC_REASON_NONE, iterations = 0, evaluations = 0;
  long sign_noise_events = 0, marginal_decisions = 0;
  double initial_cost = 0.0, final_cost = 0.0;
  std::vector<double> x;
};

class Minimizer {
 public:
  Minimizer(const Scene& s, const std::vector<Block>& blocks, std::vector<TraceRow>* trace)
      : s_(s), blocks_(blocks), P_(s.d.P), M_(s.d.M), trace_(trace) {
    lower_.assign(P_, -std::numeric_limits<double>::max());
    upper_.assign(P_, std::numeric_limits<double>::max());
    for (int b = 0; b < s.d.nbounded && b < s.d.nb; ++b) {   // src/optimizer.cpp:373-379
      lower_[2 * b] = s.prm->v_min; upper_[2 * b] = s.prm->v_max;
      lower_[2 * b + 1] = s.prm->w_min; upper_[2 * b + 1] = s.prm->w_max;
    }
  }

  // parameter_block.h Plus(): x + delta, then max(lower), then min(upper).
  void Plus(const double* x, const double* delta, double* out) const {
    for (int q = 0; q < P_; ++q) { double v = x[q] + delta[q]; v = std::max(v, lower_[q]); v = std::min(v, upper_[q]); out[q] = v; }
  }

  SolveResult Run(const double* x_init) {
    const smpc_params& prm = *s_.prm;
    SolveResult res;
    const long events0 = g_sign_noise_events, marg0 = g_marginal_decisions;
    x_.assign(x_init, x_init + P_);
    r_.assign(M_, 0.0); J_.assign(static_cast<size_t>(M_) * P_, 0.0); g_.assign(P_, 0.0); scale_.assign(P_, 1.0);
    std::vector<double> zero(P_, 0.0), cand(P_), delta(P_), step(P_);
    // IterationZero: project the start point into the box (A.4), evaluate, Jacobi scaling (A.5).
    Plus(x_.data(), zero.data(), cand.data()); x_ = cand;
    x_norm_ = Norm2(x_);
    iteration_ = 0;
    if (!EvaluateGradientAndJacobian(&res)) { res.status = SMPC_FAILURE; res.reason = SMPC_REASON_EVAL_FAILED; res.x = x_; res.initial_cost = res.final_cost = cost_; return res; }
    res.initial_cost = cost_;
    best_x_ = x_; double minimum_cost = cost_;
    double radius = 1e4, decrease_factor = 2.0;              // initial_trust_region_radius, LM strategy (A.6)
    const double max_radius = 1e16, min_radius = 1e-32, min_diag = 1e-6, max_diag = 1e32;
    bool reuse_diagonal = false;
    std::vector<double> diagonal(P_, 0.0);
    int num_consecutive_invalid = 0;
    bool step_successful = true;  // iteration 0 counts as successful
    bool at_least_one_successful = false;
    Trace(0, cost_, 0.0, 0.0, 0.0, radius, 0, 1);
    res.status = SMPC_NO_CONVERGENCE; res.reason = SMPC_REASON_MAX_ITERATIONS;
    for (;;) {
      // FinalizeIterationAndCheckIfMinimizerCanContinue
      if (step_successful && cost_ < minimum_cost) { minimum_cost = cost_; best_x_ = x_; }
      if (step_successful && iteration_ == 0) best_x_ = x_;
      if (iteration_ >= prm.max_iterations) { res.status = SMPC_NO_CONVERGENCE; res.reason = SMPC_REASON_MAX_ITERATIONS; break; }
      if (step_successful && gradient_max_norm_ <= prm.gradient_tol && !prm.fixed_iterations) { res.status = SMPC_CONVERGENCE; res.reason = SMPC_REASON_GRADIENT_TOL; break; }
      if (radius <= min_radius) { res.status = SMPC_CONVERGENCE; res.reason = SMPC_REASON_MIN_RADIUS; break; }
      ++iteration_;
      step_successful = false;

      // ComputeTrustRegionStep (LM strategy ComputeStep + model cost change, A.6-A.7)
      if (!reuse_diagonal) {
        for (int q = 0; q < P_; ++q) {
          double n2 = 0.0;
          for (int k = 0; k < M_; ++k) n2 += J_[k * P_ + q] * J_[k * P_ + q];
          diagonal[q] = std::min(std::max(n2, min_diag), max_diag);
        }
      }
      std::vector<double> lm_diag(P_);
      for (int q = 0; q < P_; ++q) lm_diag[q] = std::sqrt(diagonal[q] / radius);
      bool solved = LinearSolve(lm_diag, step.data());
      reuse_diagonal = true;
      bool step_valid = false;
      double model_cost_change = 0.0;
      if (solved) {
        bool finite = true;
        for (int q = 0; q < P_; ++q) if (!std::isfinite(step[q])) finite = false;
        if (finite) {
          for (int q = 0; q < P_; ++q) step[q] = -step[q];
          // model_cost_change = -(J step)^T (r + J step / 2)
          double acc = 0.0;
          for (int k = 0; k < M_; ++k) {
            double mr = 0.0;
            for (int q = 0; q < P_; ++q) mr += J_[k * P_ + q] * step[q];
            acc += mr * (r_[k] + mr / 2.0);
          }
          model_cost_change = -acc;
          step_valid = model_cost_change > 0.0;
        }
      }
      if (!step_valid) {
        // HandleInvalidStep
        if (++num_consecutive_invalid >= 5) { res.status = SMPC_FAILURE; res.reason = SMPC_REASON_INVALID_STEPS; break; }
        radius = radius / decrease_factor; decrease_factor *= 2.0; reuse_diagonal = false;  // StepIsInvalid
        Trace(iteration_, cost_, 0.0, 0.0, 0.0, radius, 0, 0);
        continue;
      }
      num_consecutive_invalid = 0;
      for (int q = 0; q < P_; ++q) delta[q] = step[q] * scale_[q];  // undo Jacobi scaling

      // Projected Armijo line search (problem is bounds-constrained, A.8).
      int ls_evals = 0;
      DoLineSearch(&delta, &ls_evals, &res);

      // ComputeCandidatePointAndEvaluateCost
      Plus(x_.data(), delta.data(), cand.data());
      double cand_cost;
      ++res.evaluations;
      if (!Evaluate(s_, blocks_, cand.data(), nullptr, nullptr, &cand_cost, nullptr) || !std::isfinite(cand_cost)) cand_cost = std::numeric_limits<double>::max();

      // ParameterToleranceReached
      double step_norm = 0.0;
      for (int q = 0; q < P_; ++q) step_norm += (x_[q] - cand[q]) * (x_[q] - cand[q]);
      step_norm = std::sqrt(step_norm);
      const bool tol_allowed = !prm.fixed_iterations && (!prm.tol_needs_successful_step || at_least_one_successful);
      if (tol_allowed && step_norm <= prm.param_tol * (x_norm_ + prm.param_tol)) {
        res.status = SMPC_CONVERGENCE; res.reason = SMPC_REASON_PARAMETER_TOL;
        Trace(iteration_, cost_, cost_ - cand_cost, step_norm, 0.0, radius, ls_evals, 0);
        break;
      }
      // FunctionToleranceReached
      const double cost_change = cost_ - cand_cost;
      if (tol_allowed) NoteDecision(std::fabs(cost_change), prm.fn_tol * cost_, cost_);
      if (tol_allowed && std::fabs(cost_change) <= prm.fn_tol * cost_) {
        res.status = SMPC_CONVERGENCE; res.reason = SMPC_REASON_FUNCTION_TOL;
        Trace(iteration_, cost_, cost_change, step_norm, 0.0, radius, ls_evals, 0);
        break;
      }
      // IsStepSuccessful: relative_decrease > min_relative_decrease (1e-3)
      double rho;
      if (cand_cost >= std::numeric_limits<double>::max()) rho = std::numeric_limits<double>::lowest();
      else rho = (cost_ - cand_cost) / model_cost_change;
      if (cand_cost < std::numeric_limits<double>::max()) NoteDecision(cost_ - cand_cost, 1e-3 * model_cost_change, cost_);
      if (rho > 1e-3) {
        // HandleSuccessfulStep
        x_ = cand; x_norm_ = Norm2(x_);
        if (!EvaluateGradientAndJacobian(&res)) { res.status = SMPC_FAILURE; res.reason = SMPC_REASON_EVAL_FAILED; break; }
        step_successful = true; at_least_one_successful = true;
        radius = radius / std::max(1.0 / 3.0, 1.0 - std::pow(2.0 * rho - 1.0, 3));  // StepAccepted
        radius = std::min(max_radius, radius);
        decrease_factor = 2.0; reuse_diagonal = false;
      } else {
        radius = radius / decrease_factor; decrease_factor *= 2.0; reuse_diagonal = true;  // StepRejected
      }
      Trace(iteration_, cost_, cost_change, step_norm, rho, radius, ls_evals, step_successful ? 1 : 0);
    }
    res.iterations = iteration_;
    res.x = best_x_;
    res.final_cost = minimum_cost;
    res.sign_noise_events = g_sign_noise_events - events0;
    res.marginal_decisions = g_marginal_decisions - marg0;
    return res;
  }

 private:
  static double Norm2(const std::vector<double>& v) { double s = 0.0; for (double t : v) s += t * t; return std::sqrt(s); }

  void Trace(int it, double cost, double cc, double sn, double rho, double radius, int ls, int acc) {
    if (trace_) trace_->push_back(TraceRow{static_cast<double>(it), cost, cc, gradient_max_norm_, sn, rho, radius, static_cast<double>(ls), static_cast<double>(acc)});
  }

  bool EvaluateGradientAndJacobian(SolveResult* res) {
    ++res->evaluations;
    if (!Evaluate(s_, blocks_, x_.data(), r_.data(), J_.data(), &cost_, g_.data())) return false;
    if (iteration_ == 0) {  // Jacobi scaling, computed once (A.5)
      for (int q = 0; q < P_; ++q) {
        double n2 = 0.0;
        for (int k = 0; k < M_; ++k) n2 += J_[k * P_ + q] * J_[k * P_ + q];
        scale_[q] = 1.0 / (1.0 + std::sqrt(n2));
      }
    }
    for (int k = 0; k < M_; ++k) for (int q = 0; q < P_; ++q) J_[k * P_ + q] *= scale_[q];
    // projected gradient norm: || x - Plus(x, -g) ||_inf
    std::vector<double> ng(P_), proj(P_);
    for (int q = 0; q < P_; ++q) ng[q] = -g_[q];
    Plus(x_.data(), ng.data(), proj.data());
    gradient_max_norm_ = 0.0;
    for (int q = 0; q < P_; ++q) gradient_max_norm_ = std::max(gradient_max_norm_, std::fabs(x_[q] - proj[q]));
    return true;
  }

  // Solve min ||J y - r||^2 + ||D y||^2 (scaled J). Step = -y is applied by the caller.
  bool LinearSolve(const std::vector<double>& D, double* y) {
    const int type = s_.prm->linear_solver_type;
    if (type == SMPC_DENSE_QR) {
      const int m = M_ + P_;
      std::vector<double> A(static_cast<size_t>(m) * P_, 0.0), b(m, 0.0);
      std::memcpy(A.data(), J_.data(), sizeof(double) * M_ * P_);
      for (int q = 0; q < P_; ++q) A[(M_ + q) * P_ + q] = D[q];
      std::memcpy(b.data(), r_.data(), sizeof(double) * M_);
      return HouseholderSolve(A, b, m, P_, y);
    }
    std::vector<double> H(P_ * P_, 0.0), rhs(P_, 0.0);
    for (int k = 0; k < M_; ++k) {
      const double* row = &J_[k * P_];
      for (int a = 0; a < P_; ++a) {
        if (row[a] == 0.0) continue;
        for (int b = 0; b < P_; ++b) H[a * P_ + b] += row[a] * row[b];
        rhs[a] += row[a] * r_[k];
      }
    }
    for (int q = 0; q < P_; ++q) H[q * P_ + q] += D[q] * D[q];
    if ((type == SMPC_DENSE_SCHUR || type == SMPC_SPARSE_SCHUR) && P_ > 2) {
      // Schur complement: eliminate the first 2x2 parameter block (all blocks co-occur in late-horizon
      // residuals, so the independent set has one block), Cholesky on the reduced system, back-substitute.
      const int ne = 2, nf = P_ - 2;
      std::vector<double> E{H[0], H[1], H[P_], H[P_ + 1]};
      if (!Cholesky(E, ne)) return false;
      std::vector<double> S(nf * nf), rf(nf);
      // columns of E^{-1} [H_ef | g_e]
      std::vector<double> EiHef(ne * nf);
      for (int j = 0; j < nf; ++j) { double col[2] = {H[0 * P_ + 2 + j], H[1 * P_ + 2 + j]}; CholeskySolve(E, ne, col); EiHef[0 * nf + j] = col[0]; EiHef[1 * nf + j] = col[1]; }
      double Eig[2] = {rhs[0], rhs[1]}; CholeskySolve(E, ne, Eig);
      for (int i = 0; i < nf; ++i) {
        for (int j = 0; j < nf; ++j) S[i * nf + j] = H[(2 + i) * P_ + 2 + j] - (H[(2 + i) * P_ + 0] * EiHef[0 * nf + j] + H[(2 + i) * P_ + 1] * EiHef[1 * nf + j]);
        rf[i] = rhs[2 + i] - (H[(2 + i) * P_ + 0] * Eig[0] + H[(2 + i) * P_ + 1] * Eig[1]);
      }
      if (!Cholesky(S, nf)) return false;
      CholeskySolve(S, nf, rf.data());
      double ye[2] = {rhs[0], rhs[1]};
      for (int j = 0; j < nf; ++j) { ye[0] -= H[0 * P_ + 2 + j] * rf[j]; ye[1] -= H[1 * P_ + 2 + j] * rf[j]; }
      CholeskySolve(E, ne, ye);
      y[0] = ye[0]; y[1] = ye[1];
      for (int j = 0; j < nf; ++j) y[2 + j] = rf[j];
      return true;
    }
    if (!Cholesky(H, P_)) return false;
    for (int q = 0; q < P_; ++q) y[q] = rhs[q];
    CholeskySolve(H, P_, y);
    return true;
  }

  // LineSearchFunction::Evaluate at step size a along `dir` from x_: value and directional derivative.
  void LsEvaluate(const std::vector<double>& dir, double a, FunctionSample* out, SolveResult* res) {
    *out = FunctionSample();
    out->x = a;
    std::vector<double> sd(P_), px(P_), grad(P_);
    for (int q = 0; q < P_; ++q) sd[q] = a * dir[q];
    Plus(x_.data(), sd.data(), px.data());
    double value;
    ++res->evaluations;
    const bool ok = Evaluate(s_, blocks_, px.data(), nullptr, nullptr, &value, grad.data());
    out->value = value;
    if (!ok || !std::isfinite(value)) return;
    out->value_is_valid = true;
    double gd = 0.0;
    for (int q = 0; q < P_; ++q) gd += dir[q] * grad[q];
    out->gradient = gd;
    if (!std::isfinite(gd)) return;
    out->gradient_is_valid = true;
  }

  void DoLineSearch(std::vector<double>* delta, int* n_evals, SolveResult* res) {
    const double sufficient_decrease = 1e-4, max_step_contraction = 1e-3, min_step_contraction = 0.6, min_step_size = 1e-9;
    const int max_iters = 20;
    FunctionSample initial;
    initial.x = 0.0; initial.value = cost_; initial.value_is_valid = true; initial.gradient_is_valid = true;
    double gd = 0.0;
    for (int q = 0; q < P_; ++q) gd += g_[q] * (*delta)[q];
    initial.gradient = gd;
    double dir_max = 0.0;
    for (int q = 0; q < P_; ++q) dir_max = std::max(dir_max, std::fabs((*delta)[q]));
    FunctionSample previous, current;
    LsEvaluate(*delta, 1.0, &current, res); ++*n_evals;
    int iters = 0;
    auto note = [&]() { if (current.value_is_valid) NoteDecision(current.value, initial.value + sufficient_decrease * initial.gradient * current.x, initial.value); };
    note();
    while (!current.value_is_valid || current.value > (initial.value + sufficient_decrease * initial.gradient * current.x)) {
      ++iters;
      if (iters >= max_iters) return;  // failure: delta unchanged
      double step_size;
      const double lo = max_step_contraction * current.x, hi = min_step_contraction * current.x;
      if (!current.value_is_valid) {
        step_size = std::min(std::max(current.x * 0.5, lo), hi);
      } else {
        std::vector<FunctionSample> samples;
        samples.push_back(initial);
        samples.push_back(current);
        if (previous.value_is_valid) samples.push_back(previous);
        double unused;
        MinimizeInterpolatingPolynomial(samples, lo, hi, &step_size, &unused);
      }
      if (step_size * dir_max < min_step_size) return;  // failure
      previous = current;
      LsEvaluate(*delta, step_size, &current, res); ++*n_evals;
      note();
    }
    for (int q = 0; q < P_; ++q) (*delta)[q] *= current.x;
  }

  const Scene& s_;
  const std::vector<Block>& blocks_;
  const int P_, M_;
  std::vector<TraceRow>* trace_;
  std::vector<double> lower_, upper_, x_, best_x_, r_, J_, g_, scale_;
  double cost_ = 0.0, x_norm_ = 0.0, gradient_max_norm_ = 0.0;
  int iteration_ = 0;
};

// ------------------------------------------------------------------------------------------------
// a12  Post-solve unpack (src/optimizer.cpp:390-446), with tf2 setRPY / getYaw round trips restated.
// ------------------------------------------------------------------------------------------------
inline void SetYaw(double yaw, double* qz, double* qw) { *qz = std::sin(yaw * 0.5); *qw = std::cos(yaw * 0.5); }  // setRPY(0,0,yaw)
inline double GetYaw(double qz, double qw) { return std::atan2(2.0 * (qw * qz), qw * qw - qz * qz); }             // tf2 getYaw, x=y=0

void Unpack(const Scene& s, const std::vector<double>& x, double* cmds, double* path) {
  const Dims& d = s.d;
  const int T = d.T;
  std::vector<double> vel(2 * T, 0.0);  // optim_velocities after the solve; entries 0..nb-1 are the blocks
  for (int b = 0; b < d.nb; ++b) { vel[2 * b] = x[2 * b]; vel[2 * b + 1] = x[2 * b + 1]; }
  // entries nb..T-1 hold their initial (unused) values in the reference; every one that is read below is
  // first overwritten by :390-394 because (CH-1)/bl >= CH/bl - 1.
  for (int i = d.CH / d.bl; i < T; ++i) { vel[2 * i] = vel[2 * ((d.CH - 1) / d.bl)]; vel[2 * i + 1] = vel[2 * ((d.CH - 1) / d.bl) + 1]; }
  std::vector<double> sv;
  for (int i = 0; i < d.CH; ++i) { sv.push_back(vel[2 * (i / d.bl)]); sv.push_back(vel[2 * (i / d.bl) + 1]); }   // :396-403
  for (int i = d.CH; i < T + 1; ++i) { sv.push_back(vel[2 * (i - 1)]); sv.push_back(vel[2 * (i - 1) + 1]); }    // :404-411
  if (cmds) std::memcpy(cmds, sv.data(), sizeof(double) * 2 * (T + 1));
  if (path) {
    double px = s.x0, py = s.y0, qz, qw;
    SetYaw(s.yaw0, &qz, &qw);  // evolving_poses[0].orientation, :225-226
    for (int i = 0; i < T + 1; ++i) {
      const double yaw = GetYaw(qz, qw);
      const double nx = px + sv[2 * i] * std::cos(yaw) * s.dt;      // :433-436
      const double ny = py + sv[2 * i] * std::sin(yaw) * s.dt;
      double nqz, nqw;
      SetYaw(yaw + sv[2 * i + 1] * s.dt, &nqz, &nqw);                // :437-439
      px = nx; py = ny; qz = nqz; qw = nqw;
      path[3 * i] = px; path[3 * i + 1] = py; path[3 * i + 2] = GetYaw(qz, qw);
    }
  }
}

void SolveOne(const smpc_params* prm, const smpc_scene_batch* sb, int b, smpc_result_batch* out, std::vector<TraceRow>* trace, int32_t* sign_events = nullptr, int32_t* marginal = nullptr) {
  Scene s;
  MakeScene(prm, sb, b, &s);
  const int P = s.d.P, T = s.d.T;
  std::vector<Block> blocks = BuildBlocks(s.d);
  Minimizer m(s, blocks, trace);
  SolveResult r = m.Run(sb->init_params + static_cast<size_t>(b) * P);
  if (out->params) std::memcpy(out->params + static_cast<size_t>(b) * P, r.x.data(), sizeof(double) * P);
  Unpack(s, r.x, out->cmds ? out->cmds + static_cast<size_t>(b) * (T + 1) * 2 : nullptr,
         out->path ? out->path + static_cast<size_t>(b) * (T + 1) * 3 : nullptr);
  if (out->status) out->status[b] = r.status;
  if (out->reason) out->reason[b] = r.reason;
  if (out->iterations) out->iterations[b] = r.iterations;
  if (out->evaluations) out->evaluations[b] = r.evaluations;
  if (out->initial_cost) out->initial_cost[b] = r.initial_cost;
  if (out->final_cost) out->final_cost[b] = r.final_cost;
  if (sign_events) sign_events[b] = static_cast<int32_t>(std::min<long>(r.sign_noise_events, 2147483647L));
  if (marginal) marginal[b] = static_cast<int32_t>(std::min<long>(r.marginal_decisions, 2147483647L));
}

}  // namespace

extern "C" {

int smpc_oracle_dims(const smpc_params* p, int T, int has_people, int* CH, int* bl, int* nb, int* P, int* M, int* nbounded) {
  Dims d = MakeDims(*p, T, 0, has_people != 0);
  if (CH) *CH = d.CH; if (bl) *bl = d.bl; if (nb) *nb = d.nb; if (P) *P = d.P; if (M) *M = d.M; if (nbounded) *nbounded = d.nbounded;
  return 0;
}

// Solve all scenes on `nthreads` host threads (one solve per thread at a time, like Ceres' default num_threads = 1).
int smpc_oracle_solve_batch2(const smpc_params* prm, const smpc_scene_batch* sb, smpc_result_batch* out, int nthreads, int32_t* sign_noise_events, int32_t* marginal_decisions);

int smpc_oracle_solve_batch(const smpc_params* prm, const smpc_scene_batch* sb, smpc_result_batch* out, int nthreads) {
  return smpc_oracle_solve_batch2(prm, sb, out, nthreads, nullptr, nullptr);
}

// As above; additionally sign_noise_events[B] (may be null) = per-scene count of social-force evaluations whose
// sign(theta) was decided by rounding noise (see g_sign_noise_events).
// marginal_decisions[B] (may be null) = per-scene count of decisions taken inside rounding noise (g_marginal_decisions).
int smpc_oracle_solve_batch2(const smpc_params* prm, const smpc_scene_batch* sb, smpc_result_batch* out, int nthreads, int32_t* sign_noise_events, int32_t* marginal_decisions) {
  if (!prm || !sb || !out || sb->on_device) return SMPC_ERR_INVALID_ARG;
  Dims d0 = MakeDims(*prm, sb->T, sb->N, true);
  if (d0.nb > SMPC_MAX_BLOCKS) return SMPC_ERR_UNSUPPORTED;
  nthreads = std::max(1, std::min(nthreads, sb->B));
  std::vector<std::thread> pool;
  for (int t = 0; t < nthreads; ++t) {
    pool.emplace_back([=]() { for (int b = t; b < sb->B; b += nthreads) SolveOne(prm, sb, b, out, nullptr, sign_noise_events, marginal_decisions); });
  }
  for (auto& th : pool) th.join();
  return SMPC_OK;
}

// Residuals / Jacobian / cost / gradient at given parameters (dual-number path when jacobian or gradient is asked).
int smpc_oracle_eval_batch(const smpc_params* prm, const smpc_scene_batch* sb, const double* params, smpc_eval_batch_out* out) {
  if (!prm || !sb || !out || !params || sb->on_device) return SMPC_ERR_INVALID_ARG;
  for (int b = 0; b < sb->B; ++b) {
    Scene s;
    if (!MakeScene(prm, sb, b, &s)) return SMPC_ERR_UNSUPPORTED;
    const int P = s.d.P, M = s.d.M;
    std::vector<Block> blocks = BuildBlocks(s.d);
    // M can differ per scene (has_people); outputs are strided by the batch maximum (has_people = 1).
    Dims dmax = MakeDims(*prm, sb->T, sb->N, true);
    std::vector<double> r(M), J(static_cast<size_t>(M) * P), g(P);
    double cost;
    Evaluate(s, blocks, params + static_cast<size_t>(b) * P, r.data(), (out->jacobian || out->gradient) ? J.data() : nullptr, &cost,
             (out->jacobian || out->gradient) ? g.data() : nullptr);
    if (out->residuals) { double* dst = out->residuals + static_cast<size_t>(b) * dmax.M; std::fill(dst, dst + dmax.M, 0.0); std::memcpy(dst, r.data(), sizeof(double) * M); }
    if (out->jacobian) { double* dst = out->jacobian + static_cast<size_t>(b) * dmax.M * P; std::fill(dst, dst + static_cast<size_t>(dmax.M) * P, 0.0); std::memcpy(dst, J.data(), sizeof(double) * M * P); }
    if (out->cost) out->cost[b] = cost;
    if (out->gradient) std::memcpy(out->gradient + static_cast<size_t>(b) * P, g.data(), sizeof(double) * P);
  }
  return SMPC_OK;
}

// Scalar FP64 operations of the Jet passes of ONE Jacobian evaluation of scene `scene` (the residual-only double pass
// is not instrumented). out[7] = add, mul, div, sqrt, exp, sin/cos, atan2. -1 unless built with SMPC_ORACLE_COUNT_OPS.
int smpc_oracle_count_ops(const smpc_params* prm, const smpc_scene_batch* sb, int scene, const double* params, long* out) {
#ifdef SMPC_ORACLE_COUNT_OPS
  if (!prm || !sb || !params || !out || scene < 0 || scene >= sb->B) return SMPC_ERR_INVALID_ARG;
  Scene s;
  if (!MakeScene(prm, sb, scene, &s)) return SMPC_ERR_UNSUPPORTED;
  std::vector<Block> blocks = BuildBlocks(s.d);
  std::vector<double> r(s.d.M), J(static_cast<size_t>(s.d.M) * s.d.P), g(s.d.P);
  double cost;
  g_ops = OpCount{0, 0, 0, 0, 0, 0, 0};
  Evaluate(s, blocks, params, r.data(), J.data(), &cost, g.data());
  out[0] = g_ops.add; out[1] = g_ops.mul; out[2] = g_ops.div; out[3] = g_ops.sqrt; out[4] = g_ops.exp; out[5] = g_ops.sincos; out[6] = g_ops.atan2;
  return SMPC_OK;
#else
  (void)prm; (void)sb; (void)scene; (void)params; (void)out;
  return -1;
#endif
}

// diagnostics of the polynomial root finder (single-threaded runs): out[0] = calls, out[1] = total iterations
void smpc_oracle_aberth_stats(long* out) { out[0] = g_aberth_calls; out[1] = g_aberth_iters; }

// key 1: theta := 0 when both velocities are exactly equal (see g_opt_theta_zero_when_equal_velocities).
int smpc_oracle_set_option(int key, int value) {
  if (key == 1) { g_opt_theta_zero_when_equal_velocities = value; return 0; }
  return SMPC_ERR_INVALID_ARG;
}

// Per-iteration trace of one scene: rows of 9 doubles
// [iter, cost, cost_change, gradient_max_norm, step_norm, rho, radius, line_search_evals, accepted].
int smpc_oracle_solve_trace(const smpc_params* prm, const smpc_scene_batch* sb, int scene, double* rows, int max_rows, smpc_result_batch* out1) {
  if (!prm || !sb || scene < 0 || scene >= sb->B) return SMPC_ERR_INVALID_ARG;
  std::vector<TraceRow> trace;
  // out1 buffers are indexed as a batch of one scene: shift the input batch view to `scene`.
  smpc_scene_batch one = *sb;
  const int T = sb->T, N = sb->N;
  Dims d = MakeDims(*prm, T, N, true);
  one.B = 1;
  one.pose0 += 3 * scene; one.init_params += static_cast<size_t>(d.P) * scene; one.path_pts += static_cast<size_t>(scene) * (T + 1) * 2;
  one.goal_yaw += scene; if (one.people) one.people += static_cast<size_t>(scene) * (T + 1) * 6 * N; if (one.has_people) one.has_people += scene;
  if (!one.costmap_shared) { one.costmap += static_cast<size_t>(scene) * sb->size_x * sb->size_y; one.costmap_origin += 2 * scene; }
  smpc_result_batch dummy; std::memset(&dummy, 0, sizeof(dummy));
  SolveOne(prm, &one, 0, out1 ? out1 : &dummy, &trace);
  const int n = std::min<int>(max_rows, static_cast<int>(trace.size()));
  for (int i = 0; i < n; ++i) std::memcpy(rows + 9 * i, &trace[i], sizeof(double) * 9);
  return static_cast<int>(trace.size());
}

}  // extern "C"
