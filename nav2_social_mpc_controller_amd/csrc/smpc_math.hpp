// smpc_math.hpp — f64 elementary functions of the sweep with their polynomial coefficients in SGPRs.
//
// Why not the device library's exp / atan2 / sincos: on gfx950 a 64-bit literal cannot be an operand of a VOP3
// instruction, so every coefficient of a library polynomial is materialised with two v_mov_b32 in front of its
// v_fmac_f64 (atan2: 43 moves + 12 selects around 22 fused multiply-adds; exp: 19 moves around 13) — measured in
// round 1 as "53 % of the Jacobian sweep's VALU issue is not FP64 arithmetic". Here the coefficients live in a
// table inside the kernel-argument segment (MathTab, filled by the host), are fetched by scalar loads and enter
// v_fma_f64 as SGPR operands: one VALU instruction per Horner step, no special-case selects (the callers' argument
// ranges are known: see each function).
//
// Coefficients: minimax fits derived by tools/gen_math_tables.py (Remez exchange, 60-digit arithmetic); relative
// errors of the polynomials: exp 3.4e-18, atan 4.4e-18, sin 3.8e-18, cos 4.2e-20 — below half an ulp, so the results
// are within ~1 ulp of the correctly rounded values (tests/test_math.py checks the shipped table against libm on
// the host, tests/test_gpu_math.py on the device).
#pragma once

#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)  // both passes of a HIP compilation; the plain C++ build (tests/native) takes the other branch
#define SMPC_MATH_FN __host__ __device__ inline
#define SMPC_TAB_AS __attribute__((address_space(4)))
#else
#define SMPC_MATH_FN inline
#define SMPC_TAB_AS
#endif

// The kernels of the per-tick chain around the solve (trajectorize, people filter, formatting, projection, staging,
// memory store, command selection) are short dependent chains: alone they leave the SIMDs mostly idle, next to the
// persistent solve kernel of another stream they would wait behind its wavefronts for every issue slot (measured: 5-10x
// longer). They run at the highest user wave priority instead; the solve kernel stays at 0.
#if defined(__HIP_DEVICE_COMPILE__)
#define SMPC_CHAIN_PRIORITY() __builtin_amdgcn_s_setprio(3)
#else
#define SMPC_CHAIN_PRIORITY() ((void)0)
#endif

namespace smpc {

struct MathTab {
  double exp_c[12];   // exp(r) = sum c_j r^j, |r| <= ln2 / 2 (c0 = c1 = 1)
  double atan_c[21];  // atan(t) = t * sum g_j s^j, s = t^2 in [0, 1] (g0 = 1)
  double sin_c[7];    // sin(r) = r * sum s_j z^j, z = r^2, |r| <= pi / 4 (s0 = 1)
  double cos_c[8];    // cos(r) = sum c_j z^j
  double log2e, ln2_hi, ln2_lo;
  double two_over_pi, pio2_hi, pio2_lo;
  double exp_min;     // arguments below this give exp = 0 after ldexp
  double asin_c[3];   // asin(s) = s * (1 + x (a_0 + x (a_1 + x a_2))), x = s^2 <= 5.4e-4 (atan2_unit)
};

// Nodes of atan2_unit(): entry k = 0..23 holds C_k = sqrt(1 - S_k^2), A_k = asin(S_k), S_k = k / 32 and a pad (32 bytes,
// one 16-byte and one 8-byte LDS read). The kernels that call atan2_unit() copy the table from their kernel arguments
// into LDS once per wave; a lane indexes it by the sine of its own reduced angle.
constexpr int kAtanNodes = 24, kAtanNodeStride = 4;
struct AtanNodeTab { double v[kAtanNodes * kAtanNodeStride]; };

typedef const MathTab SMPC_TAB_AS* MathTabP;

inline void fill_math_table(MathTab* t) {
  const double e[12] = {1.0, 1.0, 0.500000000000002, 0.16666666666666116, 0.041666666666478454, 0.008333333333577959, 0.0013888888955038148, 0.0001984126940757716, 2.480148244752477e-05, 2.755763483472972e-06, 2.7633795085070206e-07, 2.49931428982737e-08};
  const double a[21] = {1.0, -0.3333333333333293, 0.19999999999939716, -0.1428571428210159, 0.11111110996128326, -0.09090906842679508, 0.07692278221953736, -0.06666392328758373, 0.05880464652367159, -0.05253260147626228, 0.047215064550713934, -0.04217171992353108, 0.03660324241963838, -0.02984550791657816, 0.02191624823094173, -0.013838434268829624, 0.007154044709827802, -0.0028637504503931156, 0.0008240606399347864, -0.00015062935557186391, 1.307540239246079e-05};
  const double s[7] = {1.0, -0.16666666666666607, 0.008333333333318602, -0.00019841269827332517, 2.755731293372077e-06, -2.505064884722025e-08, 1.589082546890926e-10};
  const double c[8] = {1.0, -0.5, 0.0416666666666664, -0.001388888888885578, 2.480158728132959e-05, -2.755731246984722e-07, 2.0875505859173883e-09, -1.134972143903401e-11};
  for (int i = 0; i < 12; ++i) t->exp_c[i] = e[i];
  for (int i = 0; i < 21; ++i) t->atan_c[i] = a[i];
  for (int i = 0; i < 7; ++i) t->sin_c[i] = s[i];
  for (int i = 0; i < 8; ++i) t->cos_c[i] = c[i];
  t->exp_c[0] = 1.0; t->exp_c[1] = 1.0; t->atan_c[0] = 1.0; t->sin_c[0] = 1.0; t->cos_c[0] = 1.0; t->cos_c[1] = -0.5;
  t->log2e = 1.4426950408889634;
  t->ln2_hi = 6.93147180369123816490e-01;  // upper 32 bits of ln 2: k * ln2_hi is exact for |k| < 2^20
  t->ln2_lo = 1.90821492927058770002e-10;
  t->two_over_pi = 0.63661977236758134308;
  t->pio2_hi = 1.57079632679489655800;     // double(pi / 2)
  t->pio2_lo = 6.12323399573676603587e-17; // pi / 2 - pio2_hi
  t->exp_min = -745.2;
  t->asin_c[0] = 0.16666666666785204; t->asin_c[1] = 0.07499998898949106; t->asin_c[2] = 0.04467558115827848;
}

inline void fill_atan_nodes(AtanNodeTab* n) {
  // C_k, A_k of tools/gen_math_tables.py (60-digit arithmetic, rounded to nearest)
  const double ca[2 * kAtanNodes] = {
      1.0, 0.0,
      0.9995115994824673, 0.031255088499495154,
      0.998044963916957, 0.06254076179649139,
      0.9955957701296244, 0.09388787510751648,
      0.9921567416492215, 0.1253278311680654,
      0.9877175393299442, 0.1568928710204612,
      0.982264602843857, 0.1886163861754041,
      0.9757809372497497, 0.22053326092083333,
      0.9682458365518543, 0.25268025514207865,
      0.9596345332990055, 0.2850964402527462,
      0.9499177595981665, 0.31782370392788073,
      0.939061200082295, 0.3509073435910811,
      0.9270248108869579, 0.3843967744956391,
      0.9137619698258403, 0.4183463864434681,
      0.899218410621135, 0.4528165947449256,
      0.8833308765689106, 0.48787514754029293,
      0.8660254037844386, 0.5235987755982989,
      0.8472151069828724, 0.560075306226582,
      0.8267972847076845, 0.5974064166453502,
      0.8046495743489833, 0.6357112854013022,
      0.7806247497997998, 0.6751315329370317,
      0.7545435292281023, 0.7158380602251112,
      0.7261843774138906, 0.758040765426236,
      0.6952686081652184, 0.8020027778036185};
  for (int k = 0; k < kAtanNodes; ++k) {
    n->v[kAtanNodeStride * k] = ca[2 * k];
    n->v[kAtanNodeStride * k + 1] = ca[2 * k + 1];
    n->v[kAtanNodeStride * k + 2] = (double)k / 32.0;
    n->v[kAtanNodeStride * k + 3] = 0.0;
  }
}

// ---- hardware estimates (device) and their stand-ins for the host-side checks --------------------------------------
#if defined(__HIP_DEVICE_COMPILE__)
SMPC_MATH_FN double rcp_estimate(double x) { return __builtin_amdgcn_rcp(x); }
SMPC_MATH_FN double rsq_estimate(double x) { return __builtin_amdgcn_rsq(x); }
#else
// single-precision quality, like the worst case the refinements below are sized for
SMPC_MATH_FN double rcp_estimate(double x) { return (double)(1.0f / (float)x); }
SMPC_MATH_FN double rsq_estimate(double x) { return (double)(1.0f / sqrtf((float)x)); }
#endif

// 1 / sqrt(x) for a normal positive x: hardware estimate (measured on gfx950: relative error <= 5.3e-8, 24 bits;
// tests/test_gpu_math.py) + one third-order correction y (1 + e/2 + 3 e^2 / 8), e = 1 - x y^2 (error ~ e^3 / 3: 1e-22).
// No zero / infinity / denormal cases: the callers' arguments are squared lengths >= 1e-12. The host stand-in of the
// estimate is single precision too, so the same code is checked on the CPU.
SMPC_MATH_FN double rsqrt_pos(double x) {
  const double y = rsq_estimate(x);
  const double e = fma(-(x * y), y, 1.0);
  return fma(y * e, fma(e, 0.375, 0.5), y);
}

// a / b for normal b, |a / b| far from the overflow / underflow thresholds (no div_scale / div_fixup): one Newton step
// on the 24-bit reciprocal estimate (-> 48 bits), then one correction of the quotient (error below one ulp).
SMPC_MATH_FN double div_fast(double a, double b) {
  double r = rcp_estimate(b);
  r = fma(fma(-b, r, 1.0), r, r);
  const double q = a * r;
  return fma(fma(-b, q, a), r, q);
}

// exp(x) for x <= 700 (every caller passes a non-positive argument: -(distance / B) - (n B phi)^2, -d^2 / d0^2).
// x below -745 gives 0 (through ldexp); a NaN argument gives 0 as well (fmax drops it) — the callers' other outputs
// carry the NaN then.
SMPC_MATH_FN double exp_tab(MathTabP t, double x) {
#if defined(__HIP_DEVICE_COMPILE__)
  // one v_max_f64 (a NaN argument comes out as exp_min): the generic fmax adds two canonicalising v_max_f64
  asm("v_max_f64 %0, %1, %2" : "=v"(x) : "v"(x), "s"(t->exp_min));
#else
  x = __builtin_fmax(x, t->exp_min);
#endif
  const double kf = __builtin_rint(x * t->log2e);
  double r = fma(-kf, t->ln2_hi, x);
  r = fma(-kf, t->ln2_lo, r);
  double p = t->exp_c[11];
#pragma unroll
  for (int j = 10; j >= 2; --j) p = fma(p, r, t->exp_c[j]);
  p = fma(p, r, 1.0);
  p = fma(p, r, 1.0);
  return ldexp(p, (int)kf);
}

// atan2(y, x) for a direction vector: x^2 + y^2 within a few orders of magnitude of 1, not both zero, finite (the
// callers pass cross / dot products of unit vectors). Octant reduction to t = min / max in [0, 1], one polynomial,
// result = C + sigma * atan(t) with C in {0, pi/2, pi} (their low-order parts, <= 1.3e-16, are below half an ulp of
// the results they belong to and are dropped).
SMPC_MATH_FN double atan2_dir(MathTabP t, double y, double x) {
  const double ax = fabs(x), ay = fabs(y);
#if defined(__HIP_DEVICE_COMPILE__)
  double mx, mn;  // one instruction each (|.| as source modifiers, no canonicalising pre-pass: the inputs are finite)
  asm("v_max_f64 %0, |%1|, |%2|" : "=v"(mx) : "v"(x), "v"(y));
  asm("v_min_f64 %0, |%1|, |%2|" : "=v"(mn) : "v"(x), "v"(y));
#else
  const double mx = __builtin_fmax(ax, ay), mn = __builtin_fmin(ax, ay);
#endif
  const double q = div_fast(mn, mx);
  const double s = q * q;
  double p = t->atan_c[20];
#pragma unroll
  for (int j = 19; j >= 1; --j) p = fma(p, s, t->atan_c[j]);
  double a = fma(q * s, p, q);  // atan(q) in [0, pi/4]
  const bool swap = ay > ax, negx = x < 0.0;
  // octant: (!swap, x>=0): a | (swap, x>=0): pi/2 - a | (swap, x<0): pi/2 + a | (!swap, x<0): pi - a
  const double w = swap ? 1.0 : (negx ? 2.0 : 0.0);  // multiples of pi/2: selects on the high words of inline constants
  a = (swap != negx) ? -a : a;
  return copysign(fma(w, t->pio2_hi, a), y);
}

// atan2(y, x) of a UNIT vector (x^2 + y^2 = 1 up to rounding: the callers pass the cross / dot product of two unit
// vectors), without a division and with three polynomial steps instead of twenty. After the octant reduction the
// angle a in [0, pi/4] is known by its sine mn and its cosine mx. Node k = rint(32 mn) has sine S_k = k / 32 exactly;
// s' = mn C_k - mx S_k = sin(a - A_k) with |a - A_k| <= 0.0226, so a = A_k + asin(s') and asin(s') = s' U(s'^2) with a
// cubic U (relative error 2e-17 of a term below 0.023). Absolute error of the result: the rounding of mn C_k and of the
// tabulated C_k, A_k — about 1e-16, the size of half an ulp of a result near 1 (the results span (-pi, pi]; what the
// force needs of theta is its absolute accuracy). `nodes`: the AtanNodeTab entries (LDS in the kernels).
// Wild arguments stay inside the table (the index is clamped); a NaN argument is dropped by the min / max as in
// atan2_dir (the callers' other outputs carry the NaN).
SMPC_MATH_FN double atan2_unit(MathTabP t, const double* nodes, double y, double x) {
  const double ax = fabs(x), ay = fabs(y);
#if defined(__HIP_DEVICE_COMPILE__)
  double mx, mn;
  asm("v_max_f64 %0, |%1|, |%2|" : "=v"(mx) : "v"(x), "v"(y));
  asm("v_min_f64 %0, |%1|, |%2|" : "=v"(mn) : "v"(x), "v"(y));
#else
  const double mx = __builtin_fmax(ax, ay), mn = __builtin_fmin(ax, ay);
#endif
  // rint(32 mn) as the low word of mn + 1.5 * 2^47, whose unit in the last place is 1 / 32 (one addition, no conversion)
  const double kd = mn + 211106232532992.0;
  uint64_t kbits;
  __builtin_memcpy(&kbits, &kd, 8);
  uint32_t ki = (uint32_t)kbits;
  ki = ki < (uint32_t)(kAtanNodes - 1) ? ki : (uint32_t)(kAtanNodes - 1);
  const double* nd = nodes + kAtanNodeStride * ki;
  const double Ck = nd[0], Ak = nd[1], Sk = nd[2];
  const double sp = fma(-mx, Sk, mn * Ck);  // mx S_k enters exactly
  const double xx = sp * sp;
  double p = fma(t->asin_c[2], xx, t->asin_c[1]);
  p = fma(p, xx, t->asin_c[0]);
  double a = fma(sp * xx, p, sp) + Ak;  // in [0, pi/4]
  const bool swap = ay > ax, negx = x < 0.0;
  const double w = swap ? 1.0 : (negx ? 2.0 : 0.0);
  a = (swap != negx) ? -a : a;
  return copysign(fma(w, t->pio2_hi, a), y);
}

// sin and cos of theta for |theta| <= 1e5 (headings of a rollout: yaw0 + sum of bounded angular steps); the caller
// falls back to the library routine beyond that. Two-part Cody-Waite reduction with fused multiply-adds (the first
// product k * pio2_hi is exact inside the fma), minimax kernels on [-pi/4, pi/4], quadrant by bit tricks.
// (TabP: any pointer to a table with sin_c / cos_c / two_over_pi / pio2_hi / pio2_lo: the kernel-argument MathTab, or
// a copy a kernel keeps in registers across a loop.)
template <class TabP>
SMPC_MATH_FN void sincos_tab(TabP t, double theta, double* sn, double* cs) {
  const double kf = __builtin_rint(theta * t->two_over_pi);
  double r = fma(-kf, t->pio2_hi, theta);
  r = fma(-kf, t->pio2_lo, r);
  const double z = r * r;
  double ps = t->sin_c[6];
#pragma unroll
  for (int j = 5; j >= 1; --j) ps = fma(ps, z, t->sin_c[j]);
  const double s0 = fma(r * z, ps, r);
  double pc = t->cos_c[7];
#pragma unroll
  for (int j = 6; j >= 2; --j) pc = fma(pc, z, t->cos_c[j]);
  const double c0 = fma(z, fma(z, pc, -0.5), 1.0);
  const int n = (int)kf;
  const bool odd = (n & 1) != 0;
  double s = odd ? c0 : s0;
  double c = odd ? s0 : c0;
  // sin: negative in quadrants 2, 3; cos: negative in quadrants 1, 2
  const uint64_t sbit = (uint64_t)(uint32_t)(n & 2) << 62;
  const uint64_t cbit = (uint64_t)(uint32_t)((n + 1) & 2) << 62;
  union { double d; uint64_t u; } us, uc;
  us.d = s; uc.d = c;
  us.u ^= sbit; uc.u ^= cbit;
  *sn = us.d; *cs = uc.d;
}

}  // namespace smpc
