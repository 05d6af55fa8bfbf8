// ceres_harness.cpp — the reference's cost functors under the REAL Ceres Solver (TEST INFRASTRUCTURE ONLY).
//
// SURVEY §8(c) "escape hatch": Ceres and Eigen are absent from the image this repository is built in, so parity is
// pinned to a restated solver only (oracle/smpc_oracle.cpp). This file is the source that turns "parity unpinned"
// into a measured statement on any machine that HAS Ceres (libceres-dev, as package.xml:27 of the reference asks):
//
//     make -C oracle ceres                     # needs <ceres/ceres.h>; prints what is missing otherwise
//     python oracle/ceres_check.py             # dumps the committed golden scenes, runs the harness, compares
//
// It builds, per scene, exactly the ceres::Problem of Optimizer::optimize (reference src/optimizer.cpp:241-379):
//   * one ceres::DynamicAutoDiffCostFunction<F, 4> per (critic, step) over the SAME functor text the oracle evaluates
//     (oracle/smpc_functors.inc, instantiated here on ceres::Jet<double, 4>), parameter blocks 0..blk(i) of two doubles
//     each, residual order of :263-363;
//   * ceres::AutoDiffCostFunction<F, 1, 2, 2> for the velocity-feasibility rows (:364-370);
//   * box bounds on blocks 0..CH/bl-1 (:373-379); Solver::Options of :117-131;
//   * the costmap through ceres::Grid2D<unsigned char> + ceres::BiCubicInterpolator (:167-170), also for the residual-
//     only (double) evaluations, so no line of the restated interpolation is involved;
// solves it with ceres::Solve, and writes x, cost, iterations and the termination type per scene. It also times the
// solves: the true Ceres CPU baseline BASELINE.json asks for ("cpu_baseline.kind": "reference" once it can be built).
//
// NOT compiled in this image: oracle/Makefile's `ceres` target checks for the header first. No stand-in for Ceres,
// Eigen or glog exists in this repository.
//
// Input (written by oracle/ceres_check.py from tests/golden/*_scenes.npz), little-endian:
//   int32  B T N size_x size_y costmap_shared | double dt resolution | smpc_params (raw struct, include/smpc.h)
//   double pose0[B][3] init_params[B][P] path_pts[B][T+1][2] goal_yaw[B] people[B][T+1][6][N] | uint8 has_people[B]
//   uint8 costmap[B or 1][size_y][size_x] | double costmap_origin[B or 1][2]
// Output: per scene one text line: scene status iterations initial_cost final_cost x[0..P-1] (17 significant digits).
#if !__has_include(<ceres/ceres.h>)
#error "Ceres Solver headers not found: this harness needs libceres-dev (and Eigen). It is not built by default."
#else

#include <ceres/ceres.h>
#include <ceres/cubic_interpolation.h>
#include <ceres/dynamic_autodiff_cost_function.h>

#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <limits>
#include <memory>
#include <vector>

#include "../include/smpc.h"

// Elementary functions under the names the functor text uses, for ceres::Jet: found by argument-dependent lookup.
namespace ceres {
template <typename T, int N> inline Jet<T, N> Sqrt(const Jet<T, N>& f) { return sqrt(f); }
template <typename T, int N> inline Jet<T, N> Exp(const Jet<T, N>& f) { return exp(f); }
template <typename T, int N> inline Jet<T, N> Sin(const Jet<T, N>& f) { return sin(f); }
template <typename T, int N> inline Jet<T, N> Cos(const Jet<T, N>& f) { return cos(f); }
template <typename T, int N> inline Jet<T, N> Atan2(const Jet<T, N>& g, const Jet<T, N>& f) { return atan2(g, f); }
}  // namespace ceres

namespace harness {

using JetT = ceres::Jet<double, 4>;
using Interpolator = ceres::BiCubicInterpolator<ceres::Grid2D<unsigned char>>;

#define SMPC_OPS(field, n) ((void)0)
inline double Sqrt(double x) { return std::sqrt(x); }
inline double Exp(double x) { return std::exp(x); }
inline double Sin(double x) { return std::sin(x); }
inline double Cos(double x) { return std::cos(x); }
inline double Atan2(double y, double x) { return std::atan2(y, x); }
inline double Value(double x) { return x; }
inline double Value(const JetT& x) { return x.a; }
template <typename T> inline T MaxOf();
template <> inline double MaxOf<double>() { return std::numeric_limits<double>::max(); }
template <> inline JetT MaxOf<JetT>() { return JetT(std::numeric_limits<double>::max()); }
// declared ahead of the functor text: ceres::Jet lives in namespace ceres, so argument-dependent lookup from inside the
// templates would not find an overload of this namespace that is declared after them
inline void ZeroValue(JetT& x) { x.a = 0.0; }

#define SMPC_FUNCTORS_EXTERNAL_INTERP 1
#include "smpc_functors.inc"

// critics/obstacle_cost_function.hpp:161: costmap_interpolator_->Evaluate(row = y cell, col = x cell, &value)
inline double InterpEval(const Scene& s, double r, double c) {
  double f;
  static_cast<const Interpolator*>(s.interpolator)->Evaluate(r, c, &f);
  return f;
}
inline JetT InterpEval(const Scene& s, const JetT& r, const JetT& c) {
  JetT f;
  static_cast<const Interpolator*>(s.interpolator)->Evaluate(r, c, &f);
  return f;
}

// One per-step residual block: XCost::operator() of the reference (critics/*_cost_function.hpp), one residual.
struct StepFunctor {
  const Scene* s;
  Block b;
  template <typename T> bool operator()(T const* const* parameters, T* residuals) const {
    residuals[0] = EvalDynamic<T>(*s, b, parameters);
    return true;
  }
};

// VelocityFeasibilityCost (critics/velocity_feasibility_cost_function.hpp:46-47, 86-98): AutoDiffCostFunction<., 1, 2, 2>
struct FeasFunctor {
  const Scene* s;
  int i;
  template <typename T> bool operator()(const T* const state1, const T* const state2, T* residual) const {
    residual[0] = VelocityFeasibilityResidual<T>(*s, state1, state2, i);
    return true;
  }
};

struct Result { int status, iterations; double initial_cost, final_cost; std::vector<double> x; double seconds; };

Result SolveScene(const smpc_params& prm, const smpc_scene_batch& sb, int b) {
  Scene s;
  MakeScene(&prm, &sb, b, &s);
  // :167-170: Grid2D over getCharMap() with rows 0..sizeY, columns 0..sizeX, and its bicubic interpolator
  ceres::Grid2D<unsigned char> grid(s.costmap, 0, s.size_y, 0, s.size_x);
  Interpolator interp(grid);
  s.interpolator = &interp;
  const Dims& d = s.d;
  std::vector<double> x(sb.init_params + static_cast<size_t>(b) * d.P, sb.init_params + static_cast<size_t>(b + 1) * d.P);
  ceres::Problem problem;
  const std::vector<Block> blocks = BuildBlocks(d);  // the order of AddResidualBlock in :263-370
  for (const Block& blk : blocks) {
    if (blk.kind == kVelFeas) {  // :364-370: blocks i and i - 1
      problem.AddResidualBlock(new ceres::AutoDiffCostFunction<FeasFunctor, 1, 2, 2>(new FeasFunctor{&s, blk.i}), nullptr,
                               &x[2 * blk.i], &x[2 * (blk.i - 1)]);
      continue;
    }
    // :254-261, 272-289: the parameter blocks pushed so far = blocks 0 .. blk(i)
    const int visible = ((blk.i < d.CH) ? blk.i / d.bl : (d.CH - 1) / d.bl) + 1;
    auto* cost = new ceres::DynamicAutoDiffCostFunction<StepFunctor, 4>(new StepFunctor{&s, blk});
    std::vector<double*> pb;
    for (int j = 0; j < visible; ++j) { cost->AddParameterBlock(2); pb.push_back(&x[2 * j]); }
    cost->SetNumResiduals(1);
    problem.AddResidualBlock(cost, nullptr, pb);
  }
  for (int i = 0; i < d.nbounded; ++i) {  // :373-379
    problem.SetParameterLowerBound(&x[2 * i], 0, prm.v_min);
    problem.SetParameterUpperBound(&x[2 * i], 0, prm.v_max);
    problem.SetParameterLowerBound(&x[2 * i], 1, prm.w_min);
    problem.SetParameterUpperBound(&x[2 * i], 1, prm.w_max);
  }
  ceres::Solver::Options options;  // :117-131
  static const ceres::LinearSolverType kTypes[5] = {ceres::DENSE_SCHUR, ceres::SPARSE_SCHUR, ceres::DENSE_NORMAL_CHOLESKY,
                                                    ceres::DENSE_QR, ceres::SPARSE_NORMAL_CHOLESKY};  // optimizer.hpp:71-77
  options.linear_solver_type = kTypes[prm.linear_solver_type];
  options.max_num_iterations = prm.max_iterations;
  options.function_tolerance = prm.fn_tol;
  options.gradient_tolerance = prm.gradient_tol;
  options.parameter_tolerance = prm.param_tol;
  options.logging_type = ceres::SILENT;
  // (max_solver_time_in_seconds = max_time, :131, is 1.5-2 s: never reached by these solves; left at its default so
  // that a loaded machine cannot change a result)
  ceres::Solver::Summary summary;
  const auto t0 = std::chrono::steady_clock::now();
  ceres::Solve(options, &problem, &summary);
  Result r;
  r.seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  r.status = summary.termination_type == ceres::CONVERGENCE ? SMPC_CONVERGENCE
             : (summary.IsSolutionUsable() ? SMPC_NO_CONVERGENCE : SMPC_FAILURE);  // :384
  r.iterations = static_cast<int>(summary.iterations.size()) - 1;  // iteration 0 is the initial evaluation
  r.initial_cost = summary.initial_cost;
  r.final_cost = summary.final_cost;
  r.x = x;
  return r;
}

template <typename T> bool ReadArray(FILE* f, std::vector<T>* v, size_t n) {
  v->resize(n);
  return n == 0 || std::fread(v->data(), sizeof(T), n, f) == n;
}

}  // namespace harness

int main(int argc, char** argv) {
  using namespace harness;
  if (argc < 3) { std::fprintf(stderr, "usage: %s scenes.bin results.txt\n", argv[0]); return 2; }
  FILE* f = std::fopen(argv[1], "rb");
  if (!f) { std::perror(argv[1]); return 2; }
  int32_t hdr[6];
  double dtres[2];
  smpc_params prm;
  if (std::fread(hdr, sizeof(int32_t), 6, f) != 6 || std::fread(dtres, sizeof(double), 2, f) != 2 ||
      std::fread(&prm, sizeof(prm), 1, f) != 1) { std::fprintf(stderr, "short header\n"); return 2; }
  const int B = hdr[0], T = hdr[1], N = hdr[2], sx = hdr[3], sy = hdr[4], shared = hdr[5];
  const Dims d = MakeDims(prm, T, N, true);
  const size_t maps = shared ? 1 : B;
  std::vector<double> pose0, init, path, goal, people, origin;
  std::vector<uint8_t> has, costmap;
  bool ok = ReadArray(f, &pose0, size_t(B) * 3) && ReadArray(f, &init, size_t(B) * d.P) && ReadArray(f, &path, size_t(B) * (T + 1) * 2) &&
            ReadArray(f, &goal, size_t(B)) && ReadArray(f, &people, size_t(B) * (T + 1) * 6 * N) && ReadArray(f, &has, size_t(B)) &&
            ReadArray(f, &costmap, maps * sx * sy) && ReadArray(f, &origin, maps * 2);
  std::fclose(f);
  if (!ok) { std::fprintf(stderr, "short scene file\n"); return 2; }
  smpc_scene_batch sb;
  std::memset(&sb, 0, sizeof(sb));
  sb.B = B; sb.T = T; sb.N = N; sb.dt = dtres[0]; sb.resolution = dtres[1];
  sb.pose0 = pose0.data(); sb.init_params = init.data(); sb.path_pts = path.data(); sb.goal_yaw = goal.data();
  sb.people = people.data(); sb.has_people = has.data(); sb.costmap = costmap.data(); sb.costmap_shared = shared;
  sb.size_x = sx; sb.size_y = sy; sb.costmap_origin = origin.data();
  FILE* o = std::fopen(argv[2], "w");
  if (!o) { std::perror(argv[2]); return 2; }
  double total = 0.0;
  for (int b = 0; b < B; ++b) {
    const Result r = SolveScene(prm, sb, b);
    total += r.seconds;
    std::fprintf(o, "%d %d %d %.17g %.17g", b, r.status, r.iterations, r.initial_cost, r.final_cost);
    for (double v : r.x) std::fprintf(o, " %.17g", v);
    std::fprintf(o, "\n");
  }
  std::fclose(o);
  std::fprintf(stderr, "ceres %s: %d solves in %.3f s of ceres::Solve = %.1f solves/s on one core\n", CERES_VERSION_STRING, B, total,
               B / total);
  return 0;
}
#endif  // __has_include(<ceres/ceres.h>)
