// smpc_hip.hip — kernels + the C ABI of include/smpc.h (libsmpc_hip.so). gfx950 only, no CPU fallback.
//
// Kernels (one 64-lane wavefront per workgroup, split into 64/W scene slots; see smpc_device.hpp / smpc_lm.hpp):
//   smpc_solve_kernel<NB,W>  persistent sweep engine: whole ceres::Solve-equivalent (reference
//                            src/optimizer.cpp:241-446) per scene, LM state resident in registers / LDS for all
//                            <= max_iterations iterations, scenes pulled from a device-side queue;
//   smpc_eval_kernel<NB,W>   K1: one residual + Jacobian sweep, rows written to HBM (parity + roofline runs).
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "smpc_lm.hpp"
#include "smpc_project.hpp"
#include "smpc_format.hpp"
#include "smpc_trajectorize.hpp"
#include "smpc_path_window.hpp"

// ================================================================================================
// Host side of the C ABI
// ================================================================================================
namespace {

thread_local std::string g_last_error;

void set_error(const std::string& s) { g_last_error = s; }

#define SMPC_HIP_CHECK(expr)                                                                   \
  do {                                                                                         \
    hipError_t _e = (expr);                                                                    \
    if (_e != hipSuccess) {                                                                    \
      set_error(std::string(#expr) + ": " + hipGetErrorString(_e));                            \
      return SMPC_ERR_DEVICE;                                                                  \
    }                                                                                          \
  } while (0)

struct Dims { int CH, bl, nb, P, M, nbounded, nfeas; };

Dims make_dims(const smpc_params& p, int T, bool has_people) {
  Dims d;
  d.CH = p.control_horizon < T ? p.control_horizon : T;                 // src/optimizer.cpp:248
  d.bl = p.parameter_block_length < d.CH ? p.parameter_block_length : d.CH;  // :249
  if (d.bl < 1) d.bl = 1;
  d.nb = (d.CH - 1) / d.bl + 1;
  d.P = 2 * d.nb;
  d.nbounded = d.CH / d.bl;                                             // :373
  int nf = (d.CH / d.bl < T ? d.CH / d.bl : T) - 1;                      // :364
  d.nfeas = nf > 0 ? nf : 0;
  d.M = (has_people ? 8 : 5) * T + d.nfeas;
  return d;
}

}  // namespace

namespace smpc {
struct ProbeParams { int fn, n; const double* a; const double* b; double* o0; double* o1; MathTab mt; AtanNodeTab an; };
__global__ void smpc_math_probe_kernel(const ProbeParams) {
  const auto& k = *(const ProbeParams __attribute__((address_space(4)))*)__builtin_amdgcn_kernarg_segment_ptr();
  __shared__ double atab[kAtanTabDoubles];  // the LDS copy of the nodes, as in the sweep kernels
  for (int j = threadIdx.x; j < kAtanTabDoubles; j += blockDim.x) atab[j] = k.an.v[j];
  __syncthreads();
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= k.n) return;
  const double a = (k.fn == 7) ? 0.0 : k.a[i], b = k.b ? k.b[i] : 0.0;
  double r0 = 0.0, r1 = 0.0;
  switch (k.fn) {
    case 0: r0 = exp_tab(&k.mt, a); break;
    case 1: r0 = atan2_dir(&k.mt, a, b); break;
    case 2: sincos_tab(&k.mt, a, &r0, &r1); break;
    case 3: r0 = rsqrt_pos(a); break;
    case 4: r0 = div_fast(a, b); break;
    case 5: r0 = rcp_estimate(a); break;
    case 8: r0 = atan2_unit(&k.mt, atab, a, b); break;
    case 7: {  // a = [c4 c3 c2 c1 c0 lo hi _] per problem: the line search's bracketed root finder, trips in out1
      const double q[5] = {k.a[8 * i], k.a[8 * i + 1], k.a[8 * i + 2], k.a[8 * i + 3], k.a[8 * i + 4]};
      int trips = 0;
      r0 = bracketed_root<4>(q, k.a[8 * i + 5], k.a[8 * i + 6], &trips);
      r1 = (double)trips;
      break;
    }
    default: r0 = rsq_estimate(a); break;
  }
  k.o0[i] = r0;
  if (k.o1) k.o1[i] = r1;
}
}  // namespace smpc

namespace smpc {
// FP64 vector peak probe (SURVEY §7 asks for the denominator of the FP64 roofline to be measured, not assumed):
// eight independent v_fma_f64 chains per lane, nothing else in the loop.
__global__ __launch_bounds__(256) void smpc_fp64_peak_kernel(double* out, int iters) {
  double a0 = 1.0 + threadIdx.x * 1e-9, a1 = a0 + 1e-9, a2 = a0 + 2e-9, a3 = a0 + 3e-9;
  double a4 = a0 + 4e-9, a5 = a0 + 5e-9, a6 = a0 + 6e-9, a7 = a0 + 7e-9;
  const double m = 0.999999999, c = 1e-9;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      a0 = fma(a0, m, c); a1 = fma(a1, m, c); a2 = fma(a2, m, c); a3 = fma(a3, m, c);
      a4 = fma(a4, m, c); a5 = fma(a5, m, c); a6 = fma(a6, m, c); a7 = fma(a7, m, c);
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = ((a0 + a1) + (a2 + a3)) + ((a4 + a5) + (a6 + a7));
}
}  // namespace smpc

struct smpc_handle {
  smpc_params prm;
  int device;
  int num_cu;
  hipStream_t stream;
  hipEvent_t ev0, ev1;
  bool timed;
  int* queue;  // device-side scene queue head
  int share;   // smpc_set_solve_share: concurrent solve launches the persistent grid leaves room for
  double* stage_rec;  // staged people block of the latest call that did not bring its own (grow-only)
  double* stage_aux;
  size_t stage_rec_bytes, stage_aux_bytes;
  char* stage;        // grow-only arena for host-pointer calls (the plugin's B = 1 use): no hipMalloc per call
  size_t stage_cap;
  size_t stage_want;  // high-water mark of the calls so far
  char* pin;          // page-locked host mirror of the arena's first pin_cap bytes: the small arrays of a host-pointer call
  size_t pin_cap;     // travel in ONE copy each way instead of one pageable hipMemcpy per array (Staging, below)
};

namespace {

using KernelFn = void (*)(const smpc::KParams);

// vt: the batch carries a horizon per scene (smpc_scene_batch.T_scene): the instantiation that reads T, CH, bl of each
// scene from LDS instead of taking them as launch constants
template <int NB> KernelFn pick_w(int W, bool eval, bool vt) {
  if (vt) {
    if (W == 32) return eval ? smpc::smpc_eval_kernel<NB, 32, true> : smpc::smpc_solve_kernel<NB, 32, true>;
    return eval ? smpc::smpc_eval_kernel<NB, 64, true> : smpc::smpc_solve_kernel<NB, 64, true>;
  }
  if (W == 32) return eval ? smpc::smpc_eval_kernel<NB, 32> : smpc::smpc_solve_kernel<NB, 32>;
  return eval ? smpc::smpc_eval_kernel<NB, 64> : smpc::smpc_solve_kernel<NB, 64>;
}

KernelFn pick(int nb, int W, bool eval, bool vt = false) {
#ifdef SMPC_ONLY_NB  // development builds: one instantiation only (seconds instead of a minute to compile)
  return nb == SMPC_ONLY_NB ? pick_w<SMPC_ONLY_NB>(W, eval, vt) : nullptr;
#else
  switch (nb) {
    case 1: return pick_w<1>(W, eval, vt);
    case 2: return pick_w<2>(W, eval, vt);
    case 3: return pick_w<3>(W, eval, vt);
    case 4: return pick_w<4>(W, eval, vt);
    case 5: return pick_w<5>(W, eval, vt);
    case 6: return pick_w<6>(W, eval, vt);
    case 7: return pick_w<7>(W, eval, vt);
    case 8: return pick_w<8>(W, eval, vt);
    case 9: return pick_w<9>(W, eval, vt);
    case 10: return pick_w<10>(W, eval, vt);
    default: return nullptr;
  }
#endif
}

int validate(const smpc_handle* h, const smpc_scene_batch* sb, Dims* d) {
  if (!h || !sb) { set_error("null handle or scene batch"); return SMPC_ERR_INVALID_ARG; }
  if (sb->B < 0 || sb->T < 1 || sb->N < 0) { set_error("bad B/T/N"); return SMPC_ERR_INVALID_ARG; }
  if (h->prm.control_horizon < 1 || h->prm.parameter_block_length < 1) { set_error("control_horizon and parameter_block_length must be >= 1"); return SMPC_ERR_INVALID_ARG; }
  if (!sb->pose0 || !sb->init_params || !sb->path_pts || !sb->goal_yaw || !sb->costmap || !sb->costmap_origin) { set_error("null input array"); return SMPC_ERR_INVALID_ARG; }
  if ((sb->people_records != nullptr) != (sb->people_aux != nullptr)) { set_error("people_records and people_aux go together"); return SMPC_ERR_INVALID_ARG; }
  if (sb->N > 0 && !sb->people && !sb->people_records) { set_error("people is null with N > 0"); return SMPC_ERR_INVALID_ARG; }
  if (sb->size_x < 1 || sb->size_y < 1 || !(sb->resolution > 0.0)) { set_error("bad costmap geometry"); return SMPC_ERR_INVALID_ARG; }
  *d = make_dims(h->prm, sb->T, true);
  if (sb->T + 1 > smpc::kWave) { set_error("T + 1 > 64 rollout poses is not supported by the one-wave-per-scene mapping"); return SMPC_ERR_UNSUPPORTED; }
  if (sb->N > smpc::kWave) { set_error("N > 64 agents is not supported"); return SMPC_ERR_UNSUPPORTED; }
  if (!pick(d->nb, 64, false)) { set_error("more than SMPC_MAX_BLOCKS parameter blocks (nb must be 1..10)"); return SMPC_ERR_UNSUPPORTED; }
  return SMPC_OK;
}

void fill_kparams(const smpc_handle* h, const smpc_scene_batch* sb, const Dims& d, smpc::KParams* k) {
  std::memset(k, 0, sizeof(*k));
  k->B = sb->B; k->T = sb->T; k->N = sb->N;
  k->CH = d.CH; k->bl = d.bl; k->nb = d.nb; k->P = d.P; k->nbounded = d.nbounded; k->nfeas = d.nfeas;
  k->size_x = sb->size_x; k->size_y = sb->size_y; k->costmap_shared = sb->costmap_shared;
  k->dt = sb->dt; k->resolution = sb->resolution; k->inv_resolution = 1.0 / sb->resolution;
  k->prm = h->prm;
  k->e_M = d.M;
  k->hp_A = sb->N;  // set by launch() once the slot width is chosen
  smpc::fill_math_table(&k->mt);
  smpc::fill_atan_nodes(&k->an);
}

#define SMPC_TRY_(expr) do { int _rc = (expr); if (_rc != SMPC_OK) return _rc; } while (0)

// Host-pointer batches are staged through device memory by this helper: sub-allocations of the handle's arena, which
// grows to the high-water mark at the start of the next call (every host-pointer call ends with a stream synchronise,
// so nothing is in flight then); what does not fit meanwhile comes from hipMalloc and is freed when the call returns.
// The plugin's own call (B = 1: a dozen arrays of a few hundred bytes and one costmap) used to spend more time in its 16
// pageable hipMemcpy calls than in its kernels. The head of the arena (kPinBytes) therefore has a page-locked mirror on
// the host: inputs that land there are gathered in the mirror and cross in ONE asynchronous copy (flush_up(), before the
// first kernel of the call), outputs that land there come back in one copy and are handed out from the mirror
// (finish()). What lies beyond the mirror (large batches) is copied array by array as before.
constexpr size_t kPinBytes = 8u << 20;
struct Staging {
  smpc_handle* h;
  size_t off = 0, need = 0;
  size_t up_lo = SIZE_MAX, up_hi = 0;  // arena bytes [up_lo, up_hi) wait in the mirror for flush_up()
  struct Deferred { void* host; size_t off, bytes; };
  std::vector<Deferred> downs;         // outputs inside the mirrored range: fetched by finish()
  std::vector<void*> overflow;
  explicit Staging(smpc_handle* handle) : h(handle) {
    if (h && h->stage_want > h->stage_cap) {
      if (h->stage) (void)hipFree(h->stage);
      h->stage = nullptr; h->stage_cap = 0;
      const size_t cap = h->stage_want + h->stage_want / 4;
      void* p = nullptr;
      if (hipMalloc(&p, cap) == hipSuccess) { h->stage = static_cast<char*>(p); h->stage_cap = cap; }
    }
    if (h && h->stage) {
      const size_t want = h->stage_cap < kPinBytes ? h->stage_cap : kPinBytes;
      if (h->pin_cap < want) {
        if (h->pin) (void)hipHostFree(h->pin);
        h->pin = nullptr; h->pin_cap = 0;
        void* p = nullptr;
        if (hipHostMalloc(&p, want, hipHostMallocDefault) == hipSuccess) { h->pin = static_cast<char*>(p); h->pin_cap = want; }
      }
    }
  }
  ~Staging() {
    for (void* p : overflow) (void)hipFree(p);
    if (h && need > h->stage_want) h->stage_want = need;
  }
  int take(size_t bytes, void** out) {
    const size_t sz = (bytes + 255) & ~(size_t)255;
    need += sz;
    if (h && h->stage && off + sz <= h->stage_cap) { *out = h->stage + off; off += sz; return SMPC_OK; }
    void* p = nullptr;
    SMPC_HIP_CHECK(hipMalloc(&p, sz));
    overflow.push_back(p);
    *out = p;
    return SMPC_OK;
  }
  // offset of a device pointer inside the mirrored head of the arena, or SIZE_MAX
  size_t mirrored(const void* dev, size_t bytes) const {
    if (!h || !h->stage || !h->pin) return SIZE_MAX;
    const char* p = static_cast<const char*>(dev);
    if (p < h->stage || p + bytes > h->stage + h->pin_cap) return SIZE_MAX;
    return (size_t)(p - h->stage);
  }
  template <typename T> int up(const T* host, size_t n, const T** dev, hipStream_t st) {
    *dev = nullptr;
    if (!host || n == 0) return SMPC_OK;
    void* p = nullptr;
    SMPC_TRY_(take(n * sizeof(T), &p));
    const size_t o = mirrored(p, n * sizeof(T));
    if (o != SIZE_MAX) {
      std::memcpy(h->pin + o, host, n * sizeof(T));
      if (o < up_lo) up_lo = o;
      if (o + n * sizeof(T) > up_hi) up_hi = o + n * sizeof(T);
    } else {
      SMPC_HIP_CHECK(hipMemcpyAsync(p, host, n * sizeof(T), hipMemcpyHostToDevice, st));
    }
    *dev = static_cast<const T*>(p);
    return SMPC_OK;
  }
  // the gathered inputs cross here: every entry point calls it before its first kernel launch
  int flush_up(hipStream_t st) {
    if (up_hi > up_lo) SMPC_HIP_CHECK(hipMemcpyAsync(h->stage + up_lo, h->pin + up_lo, up_hi - up_lo, hipMemcpyHostToDevice, st));
    up_lo = SIZE_MAX; up_hi = 0;
    return SMPC_OK;
  }
  template <typename T> int out(T* host, size_t n, T** dev) {
    *dev = nullptr;
    if (!host || n == 0) return SMPC_OK;
    void* p = nullptr;
    SMPC_TRY_(take(n * sizeof(T), &p));
    *dev = static_cast<T*>(p);
    return SMPC_OK;
  }
  template <typename T> int down(T* host, const T* dev, size_t n, hipStream_t st) {
    if (!host || !dev || n == 0) return SMPC_OK;
    const size_t o = mirrored(dev, n * sizeof(T));
    if (o != SIZE_MAX) { downs.push_back({host, o, n * sizeof(T)}); return SMPC_OK; }
    SMPC_HIP_CHECK(hipMemcpyAsync(host, dev, n * sizeof(T), hipMemcpyDeviceToHost, st));
    return SMPC_OK;
  }
  // end of a host-pointer call: the deferred outputs in one copy, the stream drained, the results handed out
  int finish(hipStream_t st) {
    if (up_hi > up_lo) SMPC_TRY_(flush_up(st));  // a call that launched nothing (empty batch)
    if (!downs.empty()) {
      size_t lo = SIZE_MAX, hi = 0;
      for (const Deferred& d : downs) { if (d.off < lo) lo = d.off; if (d.off + d.bytes > hi) hi = d.off + d.bytes; }
      SMPC_HIP_CHECK(hipMemcpyAsync(h->pin + lo, h->stage + lo, hi - lo, hipMemcpyDeviceToHost, st));
    }
    SMPC_HIP_CHECK(hipStreamSynchronize(st));
    for (const Deferred& d : downs) std::memcpy(d.host, h->pin + d.off, d.bytes);
    downs.clear();
    return SMPC_OK;
  }
};

#define SMPC_TRY(expr) do { int _rc = (expr); if (_rc != SMPC_OK) return _rc; } while (0)

int grow(double** buf, size_t* have, size_t need, hipStream_t st) {
  if (need <= *have) return SMPC_OK;
  SMPC_HIP_CHECK(hipStreamSynchronize(st));  // nothing of an earlier call may still read the old buffer
  if (*buf) SMPC_HIP_CHECK(hipFree(*buf));
  *buf = nullptr; *have = 0;
  SMPC_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(buf), need));
  *have = need;
  return SMPC_OK;
}

// The staging pass: k.people (+ pose0, has_people) -> records / aux at the given device pointers.
int launch_stage(smpc_handle* h, smpc::KParams& k, double* rec, double* aux) {
  if (k.B == 0 || k.N == 0) return SMPC_OK;
  const int W = smpc::slot_width(k.T, k.N);
  const int S = smpc::kWave / W;
  const smpc::LdsLayout L = smpc::make_layout(k.T, k.N, 2, smpc::kLayoutStage, W);
  const size_t shmem = (size_t)S * L.total * sizeof(double);
  KernelFn fn = (W == 32) ? smpc::smpc_stage_kernel<32> : smpc::smpc_stage_kernel<64>;
  if (shmem > 160 * 1024) { set_error("people block does not fit the 160 KiB LDS of one CU"); return SMPC_ERR_UNSUPPORTED; }
  if (shmem > 64 * 1024) SMPC_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
  k.stage_rec = rec; k.stage_aux = aux;
  hipLaunchKernelGGL(fn, dim3((k.B + S - 1) / S), dim3(smpc::kWave), shmem, h->stream, k);
  SMPC_HIP_CHECK(hipGetLastError());
  return SMPC_OK;
}

// Makes k.people_rec / k.people_aux valid: the caller's staged block, or the library's own staging pass into the
// handle's buffers (timed separately: smpc_last_kernel_ms() reports the solve / sweep kernel alone).
int bind_people(smpc_handle* h, const smpc_scene_batch* sb, smpc::KParams& k, Staging* st) {
  if (sb->N == 0) return SMPC_OK;
  const size_t nrec = (size_t)sb->B * sb->N * sb->T * 4, naux = (size_t)sb->B * sb->T * 2;
  if (sb->people_records) {
    if (sb->on_device) { k.people_rec = sb->people_records; k.people_aux = sb->people_aux; return SMPC_OK; }
    SMPC_TRY_(st->up(sb->people_records, nrec, &k.people_rec, h->stream));
    SMPC_TRY_(st->up(sb->people_aux, naux, &k.people_aux, h->stream));
    return SMPC_OK;
  }
  SMPC_TRY_(grow(&h->stage_rec, &h->stage_rec_bytes, nrec * sizeof(double), h->stream));
  SMPC_TRY_(grow(&h->stage_aux, &h->stage_aux_bytes, naux * sizeof(double), h->stream));
  SMPC_TRY_(st->flush_up(h->stream));
  SMPC_TRY_(launch_stage(h, k, h->stage_rec, h->stage_aux));
  k.people_rec = h->stage_rec; k.people_aux = h->stage_aux;
  return SMPC_OK;
}

// Waves per CU a solve launch takes when it is sized for having the GPU to itself (launch(), below).
int lone_waves_per_cu() {
  int lone_per_cu = 8;
  if (const char* v = std::getenv("SMPC_LONE_WAVES_PER_CU")) { const int c = std::atoi(v); if (c >= 1) lone_per_cu = c; }  // experiment knob
  return lone_per_cu;
}

// Slot width of a solve launch. A shape that fits two scenes per wave (W = 32) still runs ONE scene per wave while the
// batch is small: up to one scene per SIMD (4 x CUs), two scenes in the lockstep of one wave only make each other wait
// (their LM phases differ from trip to trip; measured 4-17 % slower than a wave each, with bit-identical results); and
// where the W = 64 kernel's helper lanes pay (helper_owner_agents(): the 64 - T lanes beyond the horizon take over half
// of every step's agent list), up to the number of waves a lone launch takes anyway (8 per CU): every sweep of every
// scene is shorter then — the plugin's own call (B = 1, 8 people) 1.24 -> 1.06 ms, 1024 scenes 1.60 -> 1.13 ms, equal at
// 2048 (tools/gpu_width.py). Decided by the shape of the batch alone (B, T, N, the handle's share): a scene's result
// never depends on timing. With helper lanes it differs from the W = 32 kernel's in the last bits (the order of the
// sums over the agents).
int solve_slot_width(const smpc_handle* h, const smpc::KParams& k) {
  const int W = smpc::slot_width(k.T, k.N);
  if (W == 64) return 64;
  if (const char* v = std::getenv("SMPC_SOLVE_WIDTH")) { const int c = std::atoi(v); if (c == 32 || c == 64) return c; }  // experiment knob
  const int share = h->share > 1 ? h->share : 1;
  const bool helpers_pay = smpc::helper_owner_agents(k.T, k.N, 64) < k.N;
  const int per_cu = helpers_pay ? lone_waves_per_cu() : 4;
  return (k.B <= per_cu * h->num_cu / share) ? 64 : 32;
}

int launch(smpc_handle* h, bool eval, smpc::KParams& k) {
  const int W = eval ? smpc::slot_width(k.T, k.N) : solve_slot_width(h, k);
  const int S = smpc::kWave / W;
  KernelFn fn = pick(k.nb, W, eval, k.T_scene != nullptr);
  const smpc::LdsLayout L = smpc::make_layout(k.T, k.N, k.P, eval ? smpc::kLayoutEval : smpc::kLayoutSolve, W);
  k.hp_A = smpc::helper_owner_agents(k.T, k.N, W);
  if (std::getenv("SMPC_NO_HELPERS")) k.hp_A = k.N;  // experiment knob (the LDS layout keeps the helper regions)
  // behind the slot blocks: the feasibility rows of every slot (solve) or the row staging blocks + parked sensitivities (K1)
  const size_t extra = eval ? (size_t)smpc::eval_extra_doubles(k.T, k.P, W) : (size_t)smpc::wave_extra_doubles(k.P, W);
  // ... and behind those the wave's copy of the arctangent's node table
  const size_t shmem = ((size_t)smpc::atan_tab_offset(S * L.total, (int)extra) + smpc::kAtanTabDoubles) * sizeof(double);
  if (shmem > 160 * 1024) { set_error("scene does not fit the 160 KiB LDS of one CU"); return SMPC_ERR_UNSUPPORTED; }
  if (shmem > 64 * 1024) SMPC_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
  if (k.B == 0) return SMPC_OK;
  int grid = (k.B + S - 1) / S;
  if (!eval) {
    // persistent sweep engine: no more waves than can be resident; scenes come from the queue
    int per_cu = 0;
    SMPC_HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(fn), smpc::kWave, shmem));
    if (h->share > 1) per_cu /= h->share;  // room for the other launches of smpc_set_solve_share
    if (per_cu < 1) per_cu = 1;
    if (const char* cap = std::getenv("SMPC_MAX_WAVES_PER_CU")) {  // experiment knob: limit resident waves per CU
      const int c = std::atoi(cap);
      if (c >= 1 && c < per_cu) per_cu = c;
    }
    if (std::getenv("SMPC_DEBUG_GRID")) std::fprintf(stderr, "[smpc] solve kernel: %zu B LDS per wave, %d waves per CU\n", shmem, per_cu);
    const int resident = per_cu * h->num_cu;
    // A launch alone on the GPU finishes soonest with two waves per SIMD (two scenes per slot at the headline batch:
    // a third wave per SIMD lengthens every trip more than it shortens the queue, and the tail of long scenes grows);
    // the third wave's registers and LDS then stay free for the launches of other streams, which is where the extra
    // occupancy pays. Only a batch with many scenes per slot takes every resident wave for itself.
    const int lone_per_cu = lone_waves_per_cu();
    const int two_per_simd = lone_per_cu * h->num_cu < resident ? lone_per_cu * h->num_cu : resident;
    if (grid > two_per_simd) grid = (grid >= 4 * resident) ? resident : two_per_simd;
    k.queue = h->queue;
    k.full_gram = std::getenv("SMPC_FULL_GRAM") ? 1 : 0;  // experiment / test knob
    k.prio_step = 0;
    if (const char* v = std::getenv("SMPC_PRIO_STEP")) { const int c = std::atoi(v); if (c >= 1) k.prio_step = c; }  // experiment knob
    SMPC_HIP_CHECK(hipMemsetAsync(h->queue, 0, sizeof(int), h->stream));
  }
#ifdef SMPC_STAMPS
  {  // diagnostic build: per-wave phase cycle sums, dumped to stderr after the launch
    static unsigned long long* d_stamps = nullptr; static int cap = 0;
    if (grid > cap) { if (d_stamps) (void)hipFree(d_stamps); SMPC_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&d_stamps), (size_t)grid * 12 * sizeof(unsigned long long))); cap = grid; }
    k.stamps = d_stamps;
  }
#endif
  SMPC_HIP_CHECK(hipEventRecord(h->ev0, h->stream));
  hipLaunchKernelGGL(fn, dim3(grid), dim3(smpc::kWave), shmem, h->stream, k);
  SMPC_HIP_CHECK(hipGetLastError());
  SMPC_HIP_CHECK(hipEventRecord(h->ev1, h->stream));
  h->timed = true;
#ifdef SMPC_STAMPS
  {
    SMPC_HIP_CHECK(hipStreamSynchronize(h->stream));
    std::vector<unsigned long long> hs((size_t)grid * 12);
    SMPC_HIP_CHECK(hipMemcpy(hs.data(), k.stamps, hs.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    double tot[12] = {0};
    for (int g = 0; g < grid; ++g) for (int i = 0; i < 12; ++i) tot[i] += (double)hs[(size_t)g * 12 + i];
    double all = 0; for (int i = 0; i < 8; ++i) all += tot[i];
    static const char* names[8] = {"fetch+load_scene", "theta+sincos", "xy-loop", "agent-loop", "sens-loop", "critics+gram", "lm+output", "ls-interpolation"};
    std::fprintf(stderr, "[stamps %s] grid=%d mean cycles/wave=%.0f:", eval ? "K1" : "solve", grid, all / grid);
    for (int i = 0; i < 8; ++i) std::fprintf(stderr, " %s=%.1f%%", names[i], 100.0 * tot[i] / all);
    std::fprintf(stderr, " | rows split: people-critics=%.1f%% vel/goal/dist=%.1f%% obstacle=%.1f%% (rest of rows = feas + gram out)\n",
                 100.0 * tot[8] / all, 100.0 * tot[9] / all, 100.0 * tot[10] / all);
    if (!eval) {
      unsigned long long dbg[8] = {0};
      (void)hipMemcpyFromSymbol(dbg, HIP_SYMBOL(smpc::g_ls_dbg), sizeof(dbg));
      std::fprintf(stderr, "[ls counters, cumulative] root4 calls %llu trips %llu | root3 calls %llu trips %llu | cubic fits %llu quintic fits %llu "
                   "(monotone shortcut %llu) generic fallback %llu\n", dbg[0], dbg[1], dbg[2], dbg[3], dbg[4], dbg[5], dbg[6], dbg[7]);
    }
  }
#endif
  return SMPC_OK;
}

int bind_inputs(smpc_handle* h, const smpc_scene_batch* sb, const Dims& d, smpc::KParams* k, Staging* st) {
  const size_t B = sb->B, T = sb->T, N = sb->N;
  const size_t nmaps = sb->costmap_shared ? 1 : B;
  if (sb->on_device) {
    k->pose0 = sb->pose0; k->init_params = sb->init_params; k->path_pts = sb->path_pts; k->goal_yaw = sb->goal_yaw;
    k->people = sb->people; k->has_people = sb->has_people; k->costmap = sb->costmap; k->costmap_origin = sb->costmap_origin;
    k->T_scene = sb->T_scene;  // device array: trusted, the kernel clamps every entry into 1..T
    return SMPC_OK;
  }
  if (sb->T_scene) {
    for (size_t i = 0; i < B; ++i)
      if (sb->T_scene[i] < 1 || sb->T_scene[i] > sb->T) { set_error("T_scene entries must lie in 1..T"); return SMPC_ERR_INVALID_ARG; }
    SMPC_TRY(st->up(sb->T_scene, B, &k->T_scene, h->stream));
  }
  SMPC_TRY(st->up(sb->pose0, B * 3, &k->pose0, h->stream));
  SMPC_TRY(st->up(sb->init_params, B * d.P, &k->init_params, h->stream));
  SMPC_TRY(st->up(sb->path_pts, B * (T + 1) * 2, &k->path_pts, h->stream));
  SMPC_TRY(st->up(sb->goal_yaw, B, &k->goal_yaw, h->stream));
  if (!sb->people_records) SMPC_TRY(st->up(sb->people, B * (T + 1) * 6 * N, &k->people, h->stream));
  SMPC_TRY(st->up(sb->has_people, B, &k->has_people, h->stream));
  SMPC_TRY(st->up(sb->costmap, nmaps * (size_t)sb->size_x * sb->size_y, &k->costmap, h->stream));
  SMPC_TRY(st->up(sb->costmap_origin, nmaps * 2, &k->costmap_origin, h->stream));
  return SMPC_OK;
}

}  // namespace

extern "C" {

int smpc_abi_version(void) { return SMPC_ABI_VERSION; }

const char* smpc_last_error(void) { return g_last_error.c_str(); }

void smpc_params_default(smpc_params* p) {
  if (!p) return;
  std::memset(p, 0, sizeof(*p));
  // code defaults of OptimizerParams::get, reference src/optimizer.cpp:26-82
  p->distance_w = 3.0; p->socialwork_w = 1.0; p->velocity_w = 0.5; p->angle_w = 0.0; p->agent_angle_w = 0.5;
  p->proxemics_w = 90.0; p->velocity_feasibility_w = 0.5; p->obstacle_w = 0.0; p->goal_align_w = 0.0;
  p->control_horizon = 5; p->parameter_block_length = 5; p->max_iterations = 100;
  p->linear_solver_type = SMPC_SPARSE_NORMAL_CHOLESKY;
  p->fn_tol = 1e-7; p->gradient_tol = 1e-10; p->param_tol = 1e-15;
  // literals of Optimizer::optimize, src/optimizer.cpp:238,375-378
  p->desired_linear_vel = 0.6; p->v_min = 0.0; p->v_max = 0.6; p->w_min = -1.4; p->w_max = 1.4;
  p->fixed_iterations = 0; p->tol_needs_successful_step = 0;
}

int smpc_dims(const smpc_params* p, int T, int has_people, int* CH, int* bl, int* nb, int* P, int* M, int* n_bounded_blocks) {
  if (!p || T < 1 || p->control_horizon < 1 || p->parameter_block_length < 1) { set_error("bad arguments to smpc_dims"); return SMPC_ERR_INVALID_ARG; }
  const Dims d = make_dims(*p, T, has_people != 0);
  if (CH) *CH = d.CH;
  if (bl) *bl = d.bl;
  if (nb) *nb = d.nb;
  if (P) *P = d.P;
  if (M) *M = d.M;
  if (n_bounded_blocks) *n_bounded_blocks = d.nbounded;
  return SMPC_OK;
}

smpc_handle* smpc_create(const smpc_params* p, int device) {
  if (!p) { set_error("null params"); return nullptr; }
  if (p->linear_solver_type < SMPC_DENSE_SCHUR || p->linear_solver_type > SMPC_SPARSE_NORMAL_CHOLESKY) {
    set_error("Invalid parameter: linear_solver_type");  // same message as reference src/optimizer.cpp:44
    return nullptr;
  }
  if (p->max_iterations < 0 || p->max_iterations > SMPC_MAX_LM_ITERATIONS) {
    set_error("Invalid parameter: max_iterations (0 .. SMPC_MAX_LM_ITERATIONS)");
    return nullptr;
  }
  int count = 0;
  hipError_t e = hipGetDeviceCount(&count);
  if (e != hipSuccess || count <= 0) { set_error("no HIP device available (this library has no CPU fallback)"); return nullptr; }
  if (device < 0 || device >= count) { set_error("device index out of range"); return nullptr; }
  if (hipSetDevice(device) != hipSuccess) { set_error("hipSetDevice failed"); return nullptr; }
  smpc_handle* h = new smpc_handle();
  h->prm = *p;
  h->device = device;
  h->stream = nullptr;
  h->timed = false;
  h->queue = nullptr;
  h->share = 1;
  h->stage_rec = nullptr;
  h->stage_aux = nullptr;
  h->stage_rec_bytes = h->stage_aux_bytes = 0;
  h->stage = nullptr;
  h->stage_cap = 0;
  h->stage_want = 0;
  h->pin = nullptr;
  h->pin_cap = 0;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) != hipSuccess) { set_error("hipGetDeviceProperties failed"); delete h; return nullptr; }
  h->num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  if (hipEventCreate(&h->ev0) != hipSuccess || hipEventCreate(&h->ev1) != hipSuccess) { set_error("hipEventCreate failed"); delete h; return nullptr; }
  if (hipMalloc(reinterpret_cast<void**>(&h->queue), sizeof(int)) != hipSuccess) { set_error("hipMalloc(queue) failed"); delete h; return nullptr; }
  return h;
}

void smpc_destroy(smpc_handle* h) {
  if (!h) return;
  (void)hipEventDestroy(h->ev0);
  (void)hipEventDestroy(h->ev1);
  if (h->queue) (void)hipFree(h->queue);
  if (h->stage_rec) (void)hipFree(h->stage_rec);
  if (h->stage_aux) (void)hipFree(h->stage_aux);
  if (h->stage) (void)hipFree(h->stage);
  if (h->pin) (void)hipHostFree(h->pin);
  delete h;
}

int smpc_set_stream(smpc_handle* h, void* hip_stream) {
  if (!h) { set_error("null handle"); return SMPC_ERR_INVALID_ARG; }
  h->stream = static_cast<hipStream_t>(hip_stream);
  return SMPC_OK;
}

double smpc_last_kernel_ms(smpc_handle* h) {
  if (!h || !h->timed) return -1.0;
  if (hipEventSynchronize(h->ev1) != hipSuccess) return -1.0;
  float ms = 0.f;
  if (hipEventElapsedTime(&ms, h->ev0, h->ev1) != hipSuccess) return -1.0;
  return (double)ms;
}

int smpc_set_solve_share(smpc_handle* h, int32_t n) {
  if (!h || n < 1) { set_error("null handle or share < 1"); return SMPC_ERR_INVALID_ARG; }
  h->share = n;
  return SMPC_OK;
}

int smpc_solve_slot_width(const smpc_handle* h, int32_t B, int32_t T, int32_t N) {
  if (!h || B < 0 || T < 1 || N < 0) { set_error("null handle or bad B/T/N"); return SMPC_ERR_INVALID_ARG; }
  if (T + 1 > smpc::kWave || N > smpc::kWave) { set_error("T + 1 > 64 rollout poses or N > 64 agents"); return SMPC_ERR_UNSUPPORTED; }
  smpc::KParams k;
  k.B = B; k.T = T; k.N = N;
  return solve_slot_width(h, k);
}

int smpc_solve_batch(smpc_handle* h, const smpc_scene_batch* sb, smpc_result_batch* out) {
  Dims d;
  SMPC_TRY(validate(h, sb, &d));
  if (!out) { set_error("null result batch"); return SMPC_ERR_INVALID_ARG; }
  SMPC_HIP_CHECK(hipSetDevice(h->device));
  smpc::KParams k;
  fill_kparams(h, sb, d, &k);
  Staging st(h);
  SMPC_TRY(bind_inputs(h, sb, d, &k, &st));
  SMPC_TRY(bind_people(h, sb, k, &st));
  const size_t B = sb->B, T = sb->T;
  if (sb->order) {  // queue order hint
    if (sb->on_device) {
      k.order = sb->order;
    } else {
      std::vector<uint8_t> seen(B, 0);
      for (size_t i = 0; i < B; ++i) {
        const int32_t v = sb->order[i];
        if (v < 0 || (size_t)v >= B || seen[v]) { set_error("order is not a permutation of 0..B-1"); return SMPC_ERR_INVALID_ARG; }
        seen[v] = 1;
      }
      SMPC_TRY(st.up(sb->order, B, &k.order, h->stream));
    }
  }
  if (sb->on_device) {
    k.o_params = out->params; k.o_cmds = out->cmds; k.o_path = out->path; k.o_status = out->status; k.o_reason = out->reason;
    k.o_iterations = out->iterations; k.o_evaluations = out->evaluations; k.o_initial_cost = out->initial_cost; k.o_final_cost = out->final_cost;
    // a device-side order cannot be checked here: should it not be a permutation, the scenes it leaves out must not keep
    // the status of an earlier call — every status starts as SMPC_NOT_SOLVED (-1) and is overwritten by the scene's solve
    if (k.order && k.o_status && sb->B > 0) SMPC_HIP_CHECK(hipMemsetAsync(k.o_status, 0xFF, (size_t)sb->B * sizeof(int32_t), h->stream));
    return launch(h, false, k);
  }
  SMPC_TRY(st.out(out->params, B * d.P, &k.o_params));
  SMPC_TRY(st.out(out->cmds, B * (T + 1) * 2, &k.o_cmds));
  SMPC_TRY(st.out(out->path, B * (T + 1) * 3, &k.o_path));
  SMPC_TRY(st.out(out->status, B, &k.o_status));
  SMPC_TRY(st.out(out->reason, B, &k.o_reason));
  SMPC_TRY(st.out(out->iterations, B, &k.o_iterations));
  SMPC_TRY(st.out(out->evaluations, B, &k.o_evaluations));
  SMPC_TRY(st.out(out->initial_cost, B, &k.o_initial_cost));
  SMPC_TRY(st.out(out->final_cost, B, &k.o_final_cost));
  SMPC_TRY(st.flush_up(h->stream));
  SMPC_TRY(launch(h, false, k));
  SMPC_TRY(st.down(out->params, k.o_params, B * d.P, h->stream));
  SMPC_TRY(st.down(out->cmds, k.o_cmds, B * (T + 1) * 2, h->stream));
  SMPC_TRY(st.down(out->path, k.o_path, B * (T + 1) * 3, h->stream));
  SMPC_TRY(st.down(out->status, k.o_status, B, h->stream));
  SMPC_TRY(st.down(out->reason, k.o_reason, B, h->stream));
  SMPC_TRY(st.down(out->iterations, k.o_iterations, B, h->stream));
  SMPC_TRY(st.down(out->evaluations, k.o_evaluations, B, h->stream));
  SMPC_TRY(st.down(out->initial_cost, k.o_initial_cost, B, h->stream));
  SMPC_TRY(st.down(out->final_cost, k.o_final_cost, B, h->stream));
  SMPC_TRY(st.finish(h->stream));
  return SMPC_OK;
}

int smpc_project_people_batch(smpc_handle* h, const smpc_projection_batch* in, double* people_proj, int32_t* error) {
  if (!h || !in || !people_proj) { set_error("null handle / input / output"); return SMPC_ERR_INVALID_ARG; }
  if (in->B < 0 || in->T < 1 || in->N < 1) { set_error("bad B/T/N"); return SMPC_ERR_INVALID_ARG; }
  if (in->N + 1 > smpc::kWave) { set_error("N + 1 > 64 agents is not supported"); return SMPC_ERR_UNSUPPORTED; }
  if (!in->init_people || !in->robot_path || !in->od_origin) { set_error("null input array"); return SMPC_ERR_INVALID_ARG; }
  // the reference throws for an empty / malformed ObstacleDistance grid (src/optimizer.cpp:676-687)
  if (!in->od_indexes) { set_error("ObstacleDistance grid is empty"); return SMPC_ERR_INVALID_ARG; }
  if (in->od_width <= 0 || in->od_height <= 0) { set_error("ObstacleDistance grid has invalid size"); return SMPC_ERR_INVALID_ARG; }
  if (!(in->od_resolution > 0.0f)) { set_error("ObstacleDistance grid has invalid resolution"); return SMPC_ERR_INVALID_ARG; }
  SMPC_HIP_CHECK(hipSetDevice(h->device));
  smpc::ProjParams p;
  std::memset(&p, 0, sizeof(p));
  p.B = in->B; p.T = in->T; p.N = in->N;
  int G = 2;
  while (G < in->N + 1) G *= 2;
  p.G = G;
  p.H = (2 * G <= smpc::kWave) ? 2 : 1;  // two copies of a scene split the partner rounds while they fit in the wavefront
  smpc::fill_math_table(&p.mt);
  p.max_time = in->max_time; p.time_step = in->time_step; p.od_resolution = in->od_resolution;
  p.od_shared = in->od_shared; p.od_width = in->od_width; p.od_height = in->od_height;
  const size_t B = in->B, T = in->T, N = in->N;
  const size_t ngrid = in->od_shared ? 1 : B;
  Staging st(h);
  if (in->on_device) {
    p.init_people = in->init_people; p.robot_path = in->robot_path; p.od_indexes = in->od_indexes; p.od_origin = in->od_origin;
    p.people_proj = people_proj; p.error = error;
  } else {
    SMPC_TRY(st.up(in->init_people, B * N * 6, &p.init_people, h->stream));
    SMPC_TRY(st.up(in->robot_path, B * (T + 1) * 6, &p.robot_path, h->stream));
    SMPC_TRY(st.up(in->od_indexes, ngrid * (size_t)in->od_width * in->od_height, &p.od_indexes, h->stream));
    SMPC_TRY(st.up(in->od_origin, ngrid * 2, &p.od_origin, h->stream));
    SMPC_TRY(st.out(people_proj, B * (T + 1) * 6 * N, &p.people_proj));
    SMPC_TRY(st.out(error, B, &p.error));
  }
  if (B > 0) {
    const int per_wave = smpc::kWave / (G * p.H);
    const int grid = (int)((B + per_wave - 1) / per_wave);
    SMPC_HIP_CHECK(hipEventRecord(h->ev0, h->stream));
    SMPC_TRY(st.flush_up(h->stream));
    hipLaunchKernelGGL(smpc::smpc_project_kernel, dim3(grid), dim3(smpc::kWave), 0, h->stream, p);
    SMPC_HIP_CHECK(hipGetLastError());
    SMPC_HIP_CHECK(hipEventRecord(h->ev1, h->stream));
    h->timed = true;
  }
  if (!in->on_device) {
    SMPC_TRY(st.down(people_proj, p.people_proj, B * (T + 1) * 6 * N, h->stream));
    SMPC_TRY(st.down(error, p.error, B, h->stream));
    SMPC_TRY(st.finish(h->stream));
  }
  return SMPC_OK;
}

int smpc_people_to_status_batch(smpc_handle* h, const smpc_people_batch* in, double* init_people, uint8_t* has_people) {
  if (!h || !in || !init_people) { set_error("null handle / input / output"); return SMPC_ERR_INVALID_ARG; }
  if (in->B < 0 || in->Np < 1 || in->N < 1) { set_error("bad B / Np / N"); return SMPC_ERR_INVALID_ARG; }
  if (!in->people || !in->count) { set_error("null input array"); return SMPC_ERR_INVALID_ARG; }
  SMPC_HIP_CHECK(hipSetDevice(h->device));
  const size_t B = in->B, Np = in->Np, N = in->N;
  smpc::PeopleParams p;
  std::memset(&p, 0, sizeof(p));
  p.B = in->B; p.Np = in->Np; p.N = in->N;
  const bool filter = in->robot_pose != nullptr;
  if (filter && (!in->costmap_origin || in->size_x < 1 || in->size_y < 1 || !(in->resolution > 0.0))) {
    set_error("field-of-view filter needs the costmap geometry"); return SMPC_ERR_INVALID_ARG;
  }
  p.fov_angle = in->fov_angle; p.costmap_shared = in->costmap_shared; p.size_x = in->size_x; p.size_y = in->size_y;
  p.resolution = in->resolution;
  Staging st(h);
  if (in->on_device) {
    p.people = in->people; p.count = in->count; p.init_people = init_people; p.has_people = has_people;
    p.robot_pose = in->robot_pose; p.costmap_origin = in->costmap_origin;
  } else {
    SMPC_TRY(st.up(in->people, B * Np * 5, &p.people, h->stream));
    SMPC_TRY(st.up(in->count, B, &p.count, h->stream));
    if (filter) {
      SMPC_TRY(st.up(in->robot_pose, B * 3, &p.robot_pose, h->stream));
      SMPC_TRY(st.up(in->costmap_origin, (in->costmap_shared ? 1 : B) * 2, &p.costmap_origin, h->stream));
    }
    SMPC_TRY(st.out(init_people, B * N * 6, &p.init_people));
    SMPC_TRY(st.out(has_people, B, &p.has_people));
  }
  if (B > 0) {
    SMPC_HIP_CHECK(hipEventRecord(h->ev0, h->stream));
    SMPC_TRY(st.flush_up(h->stream));
    hipLaunchKernelGGL(smpc::smpc_people_to_status_kernel, dim3((unsigned)((B + 255) / 256)), dim3(256), 0, h->stream, p);
    SMPC_HIP_CHECK(hipGetLastError());
    SMPC_HIP_CHECK(hipEventRecord(h->ev1, h->stream));
    h->timed = true;
  }
  if (!in->on_device) {
    SMPC_TRY(st.down(init_people, p.init_people, B * N * 6, h->stream));
    SMPC_TRY(st.down(has_people, p.has_people, B, h->stream));
    SMPC_TRY(st.finish(h->stream));
  }
  return SMPC_OK;
}

int smpc_format_to_optimize_batch(smpc_handle* h, const smpc_format_batch* in, smpc_format_out* out) {
  if (!h || !in || !out) { set_error("null handle / input / output"); return SMPC_ERR_INVALID_ARG; }
  if (in->B < 0 || in->T < 1) { set_error("bad B/T"); return SMPC_ERR_INVALID_ARG; }
  if (!in->path || !in->cmds || !in->speed || !in->memory.prev_path || !in->memory.prev_cmds || !in->memory.valid) {
    set_error("null input array / memory record"); return SMPC_ERR_INVALID_ARG;
  }
  if (!out->robot_status || !out->pose0 || !out->init_params || !out->path_pts || !out->goal_yaw) {
    set_error("null output array"); return SMPC_ERR_INVALID_ARG;
  }
  SMPC_HIP_CHECK(hipSetDevice(h->device));
  const Dims d = make_dims(h->prm, in->T, true);
  const size_t B = in->B, Tp = (size_t)in->T + 1;
  smpc::FormatParams p;
  std::memset(&p, 0, sizeof(p));
  const size_t rows = in->path_rows > 0 ? (size_t)in->path_rows : Tp;
  if (rows < Tp) { set_error("path_rows < T + 1"); return SMPC_ERR_INVALID_ARG; }
  p.B = in->B; p.T = in->T; p.nb = d.nb; p.P = d.P; p.rows = (int)rows;
  p.max_poses = in->max_poses > 0 ? in->max_poses : 0;
  p.time_step = in->time_step; p.current_path_w = in->current_path_w; p.current_cmds_w = in->current_cmds_w;
  if (in->n_poses && !in->memory.length) {
    set_error("n_poses needs memory.length: records of scenes with horizons of their own differ in size"); return SMPC_ERR_INVALID_ARG;
  }
  Staging st(h);
  if (in->on_device) {
    p.path = in->path; p.cmds = in->cmds; p.speed = in->speed; p.n_poses = in->n_poses;
    p.prev_path = in->memory.prev_path; p.prev_cmds = in->memory.prev_cmds; p.valid = in->memory.valid; p.length = in->memory.length;
    p.robot_status = out->robot_status; p.pose0 = out->pose0; p.init_params = out->init_params;
    p.path_pts = out->path_pts; p.goal_yaw = out->goal_yaw; p.T_scene = out->T_scene;
  } else {
    const double* c = nullptr; const int32_t* ci = nullptr;
    SMPC_TRY(st.up(in->n_poses, B, &p.n_poses, h->stream));
    SMPC_TRY(st.up(static_cast<const int32_t*>(in->memory.length), B * 2, &ci, h->stream)); p.length = const_cast<int32_t*>(ci);
    SMPC_TRY(st.out(out->T_scene, B, &p.T_scene));
    SMPC_TRY(st.up(in->path, B * rows * 3, &p.path, h->stream));
    SMPC_TRY(st.up(in->cmds, B * rows * 2, &p.cmds, h->stream));
    SMPC_TRY(st.up(in->speed, B * 2, &p.speed, h->stream));
    SMPC_TRY(st.up(static_cast<const double*>(in->memory.prev_path), B * Tp * 3, &c, h->stream)); p.prev_path = const_cast<double*>(c);
    SMPC_TRY(st.up(static_cast<const double*>(in->memory.prev_cmds), B * Tp * 2, &c, h->stream)); p.prev_cmds = const_cast<double*>(c);
    SMPC_TRY(st.up(static_cast<const int32_t*>(in->memory.valid), B, &ci, h->stream)); p.valid = const_cast<int32_t*>(ci);
    SMPC_TRY(st.out(out->robot_status, B * Tp * 6, &p.robot_status));
    SMPC_TRY(st.out(out->pose0, B * 3, &p.pose0));
    SMPC_TRY(st.out(out->init_params, B * (size_t)d.P, &p.init_params));
    SMPC_TRY(st.out(out->path_pts, B * Tp * 2, &p.path_pts));
    SMPC_TRY(st.out(out->goal_yaw, B, &p.goal_yaw));
  }
  if (B > 0) {
    const long long n = (long long)B * (long long)Tp;
    SMPC_HIP_CHECK(hipEventRecord(h->ev0, h->stream));
    SMPC_TRY(st.flush_up(h->stream));
    hipLaunchKernelGGL(smpc::smpc_format_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, p);
    SMPC_HIP_CHECK(hipGetLastError());
    hipLaunchKernelGGL(smpc::smpc_format_mark_kernel, dim3((unsigned)((B + 255) / 256)), dim3(256), 0, h->stream, p);
    SMPC_HIP_CHECK(hipGetLastError());
    SMPC_HIP_CHECK(hipEventRecord(h->ev1, h->stream));
    h->timed = true;
  }
  if (!in->on_device) {
    SMPC_TRY(st.down(out->robot_status, p.robot_status, B * Tp * 6, h->stream));
    SMPC_TRY(st.down(out->pose0, p.pose0, B * 3, h->stream));
    SMPC_TRY(st.down(out->init_params, p.init_params, B * (size_t)d.P, h->stream));
    SMPC_TRY(st.down(out->path_pts, p.path_pts, B * Tp * 2, h->stream));
    SMPC_TRY(st.down(out->goal_yaw, p.goal_yaw, B, h->stream));
    SMPC_TRY(st.down(in->memory.prev_path, p.prev_path, B * Tp * 3, h->stream));
    SMPC_TRY(st.down(in->memory.prev_cmds, p.prev_cmds, B * Tp * 2, h->stream));
    SMPC_TRY(st.down(in->memory.valid, p.valid, B, h->stream));
    SMPC_TRY(st.down(in->memory.length, p.length, B * 2, h->stream));
    SMPC_TRY(st.down(out->T_scene, p.T_scene, B, h->stream));
    SMPC_TRY(st.finish(h->stream));
  }
  return SMPC_OK;
}

int smpc_memory_store_batch(smpc_handle* h, int32_t B_, int32_t T, int32_t on_device, const int32_t* status,
                            const double* path, const double* cmds, smpc_memory_batch* memory, const int32_t* T_scene) {
  if (!h || !status || !path || !cmds || !memory || !memory->prev_path || !memory->prev_cmds || !memory->valid) {
    set_error("null handle / array / memory record"); return SMPC_ERR_INVALID_ARG;
  }
  if (B_ < 0 || T < 1) { set_error("bad B/T"); return SMPC_ERR_INVALID_ARG; }
  SMPC_HIP_CHECK(hipSetDevice(h->device));
  const size_t B = B_, Tp = (size_t)T + 1;
  smpc::StoreParams p;
  std::memset(&p, 0, sizeof(p));
  p.B = B_; p.T = T;
  Staging st(h);
  if (T_scene && !memory->length) {
    set_error("T_scene needs memory.length: records of scenes with horizons of their own differ in size"); return SMPC_ERR_INVALID_ARG;
  }
  if (on_device) {
    p.status = status; p.path = path; p.cmds = cmds; p.T_scene = T_scene;
    p.prev_path = memory->prev_path; p.prev_cmds = memory->prev_cmds; p.valid = memory->valid; p.length = memory->length;
  } else {
    const double* c = nullptr; const int32_t* ci = nullptr;
    SMPC_TRY(st.up(T_scene, B, &p.T_scene, h->stream));
    SMPC_TRY(st.up(static_cast<const int32_t*>(memory->length), B * 2, &ci, h->stream)); p.length = const_cast<int32_t*>(ci);
    SMPC_TRY(st.up(status, B, &p.status, h->stream));
    SMPC_TRY(st.up(path, B * Tp * 3, &p.path, h->stream));
    SMPC_TRY(st.up(cmds, B * Tp * 2, &p.cmds, h->stream));
    SMPC_TRY(st.up(static_cast<const double*>(memory->prev_path), B * Tp * 3, &c, h->stream)); p.prev_path = const_cast<double*>(c);
    SMPC_TRY(st.up(static_cast<const double*>(memory->prev_cmds), B * Tp * 2, &c, h->stream)); p.prev_cmds = const_cast<double*>(c);
    SMPC_TRY(st.up(static_cast<const int32_t*>(memory->valid), B, &ci, h->stream)); p.valid = const_cast<int32_t*>(ci);
  }
  if (B > 0) {
    const long long n = (long long)B * (long long)Tp;
    SMPC_HIP_CHECK(hipEventRecord(h->ev0, h->stream));
    SMPC_TRY(st.flush_up(h->stream));
    hipLaunchKernelGGL(smpc::smpc_memory_store_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, p);
    SMPC_HIP_CHECK(hipGetLastError());
    SMPC_HIP_CHECK(hipEventRecord(h->ev1, h->stream));
    h->timed = true;
  }
  if (!on_device) {
    SMPC_TRY(st.down(memory->prev_path, p.prev_path, B * Tp * 3, h->stream));
    SMPC_TRY(st.down(memory->prev_cmds, p.prev_cmds, B * Tp * 2, h->stream));
    SMPC_TRY(st.down(memory->valid, p.valid, B, h->stream));
    SMPC_TRY(st.down(memory->length, p.length, B * 2, h->stream));
    SMPC_TRY(st.finish(h->stream));
  }
  return SMPC_OK;
}

int smpc_trajectorize_path_batch(smpc_handle* h, const smpc_trajectorize_batch* in, smpc_trajectorize_out* out) {
  if (!h || !in || !out) { set_error("null handle / input / output"); return SMPC_ERR_INVALID_ARG; }
  if (in->B < 0 || in->L < 1 || in->max_steps < 0) { set_error("bad B / L / max_steps"); return SMPC_ERR_INVALID_ARG; }
  if (!in->plan || !in->plan_len || !in->robot_pose) { set_error("null input array"); return SMPC_ERR_INVALID_ARG; }
  if (!out->path || !out->cmds || !out->n_poses) { set_error("null output array"); return SMPC_ERR_INVALID_ARG; }
  SMPC_HIP_CHECK(hipSetDevice(h->device));
  const size_t B = in->B, L = in->L, S1 = (size_t)in->max_steps + 1;
  smpc::TrajParams p;
  std::memset(&p, 0, sizeof(p));
  p.B = in->B; p.L = in->L; p.max_steps = in->max_steps; p.omnidirectional = in->omnidirectional;
  p.desired_linear_vel = in->desired_linear_vel; p.lookahead_dist = in->lookahead_dist;
  p.max_angular_vel = in->max_angular_vel; p.time_step = in->time_step;
  smpc::fill_math_table(&p.mt);
  Staging st(h);
  if (in->on_device) {
    p.plan = in->plan; p.plan_len = in->plan_len; p.robot_pose = in->robot_pose;
    p.path = out->path; p.cmds = out->cmds; p.cmds_vy = out->cmds_vy; p.n_poses = out->n_poses; p.error = out->error;
  } else {
    SMPC_TRY(st.up(in->plan, B * L * 2, &p.plan, h->stream));
    SMPC_TRY(st.up(in->plan_len, B, &p.plan_len, h->stream));
    SMPC_TRY(st.up(in->robot_pose, B * 3, &p.robot_pose, h->stream));
    SMPC_TRY(st.out(out->path, B * S1 * 3, &p.path));
    SMPC_TRY(st.out(out->cmds, B * S1 * 2, &p.cmds));
    SMPC_TRY(st.out(out->cmds_vy, B * S1, &p.cmds_vy));
    SMPC_TRY(st.out(out->n_poses, B, &p.n_poses));
    SMPC_TRY(st.out(out->error, B, &p.error));
  }
  if (B > 0) {
    const int per_wave = smpc::kWave / smpc::kTrajGroup;
    // plans of up to 512 poses stay in registers (kR poses per lane of a 16-lane group); the raw step outputs are parked
    // in LDS (48 bytes per step and plan): four-wavefront blocks while their park fits in 48 KB (64 steps), else
    // one-wavefront blocks (up to 256 steps). Longer plans / horizons take the kernel that searches the plan in memory.
    const size_t park_step = (size_t)smpc::kTrajParkDoubles * sizeof(double);
    const size_t park_wave = (size_t)per_wave * in->max_steps * park_step;
    const int threads = (park_wave * (smpc::kTrajBlock / smpc::kWave) <= 48 * 1024) ? smpc::kTrajBlock : smpc::kWave;
    const int per_block = threads / smpc::kTrajGroup;
    const size_t park = park_wave * (threads / smpc::kWave);
    const dim3 grid((unsigned)((B + per_block - 1) / per_block)), block(threads);
    const int need = (int)((L + smpc::kTrajGroup - 1) / smpc::kTrajGroup);
    SMPC_TRY(st.flush_up(h->stream));
    SMPC_HIP_CHECK(hipEventRecord(h->ev0, h->stream));
    // plans over 512 poses: one-wavefront blocks of the 8-slot kernel with the reachable poses compacted into LDS
    const size_t list_wave = (size_t)per_wave * 8 * smpc::kTrajGroup * 2 * sizeof(double);
    if (need > 32 && park_wave + list_wave <= 48 * 1024) {
      p.compact = 1;
      hipLaunchKernelGGL(smpc::smpc_trajectorize_kernel<8>, dim3((unsigned)((B + per_wave - 1) / per_wave)), dim3(smpc::kWave),
                         park_wave + list_wave, h->stream, p);
    } else if (need > 32 || park > 48 * 1024) {
      hipLaunchKernelGGL(smpc::smpc_trajectorize_long_kernel, dim3((unsigned)((B + per_wave - 1) / per_wave)), dim3(smpc::kWave), 0, h->stream, p);
    } else if (need <= 8) {
      hipLaunchKernelGGL(smpc::smpc_trajectorize_kernel<8>, grid, block, park, h->stream, p);
    } else if (need <= 16) {
      hipLaunchKernelGGL(smpc::smpc_trajectorize_kernel<16>, grid, block, park, h->stream, p);
    } else if (need <= 25) {
      hipLaunchKernelGGL(smpc::smpc_trajectorize_kernel<25>, grid, block, park, h->stream, p);
    } else {
      hipLaunchKernelGGL(smpc::smpc_trajectorize_kernel<32>, grid, block, park, h->stream, p);
    }
    SMPC_HIP_CHECK(hipGetLastError());
    SMPC_HIP_CHECK(hipEventRecord(h->ev1, h->stream));
    h->timed = true;
  }
  if (!in->on_device) {
    SMPC_TRY(st.down(out->path, p.path, B * S1 * 3, h->stream));
    SMPC_TRY(st.down(out->cmds, p.cmds, B * S1 * 2, h->stream));
    SMPC_TRY(st.down(out->cmds_vy, p.cmds_vy, B * S1, h->stream));
    SMPC_TRY(st.down(out->n_poses, p.n_poses, B, h->stream));
    SMPC_TRY(st.down(out->error, p.error, B, h->stream));
    SMPC_TRY(st.finish(h->stream));
  }
  return SMPC_OK;
}

int smpc_transform_global_plan_batch(smpc_handle* h, const smpc_plan_window_batch* in, double* window, int32_t* window_len,
                                     int32_t* error) {
  if (!h || !in || !window || !window_len) { set_error("null handle / input / output"); return SMPC_ERR_INVALID_ARG; }
  if (in->B < 0 || in->L < 1) { set_error("bad B / L"); return SMPC_ERR_INVALID_ARG; }
  if (!in->plan || !in->plan_len || !in->plan_start || !in->robot_pose) { set_error("null input array"); return SMPC_ERR_INVALID_ARG; }
  SMPC_HIP_CHECK(hipSetDevice(h->device));
  const size_t B = in->B, L = in->L;
  smpc::WindowParams p;
  std::memset(&p, 0, sizeof(p));
  p.B = in->B; p.L = in->L; p.search_dist = in->max_robot_pose_search_dist; p.dist_threshold = in->dist_threshold;
  Staging st(h);
  if (in->on_device) {
    p.plan = in->plan; p.plan_len = in->plan_len; p.plan_start = in->plan_start; p.robot_pose = in->robot_pose;
    p.to_local = in->to_local; p.window = window; p.window_len = window_len; p.error = error;
  } else {
    const int32_t* start_in = nullptr;
    SMPC_TRY(st.up(in->plan, B * L * 2, &p.plan, h->stream));
    SMPC_TRY(st.up(in->plan_len, B, &p.plan_len, h->stream));
    SMPC_TRY(st.up(static_cast<const int32_t*>(in->plan_start), B, &start_in, h->stream));
    p.plan_start = const_cast<int32_t*>(start_in);  // device copy: read, updated in place, copied back below
    SMPC_TRY(st.up(in->robot_pose, B * 3, &p.robot_pose, h->stream));
    if (in->to_local) SMPC_TRY(st.up(in->to_local, B * 3, &p.to_local, h->stream));
    SMPC_TRY(st.out(window, B * L * 2, &p.window));
    SMPC_TRY(st.out(window_len, B, &p.window_len));
    SMPC_TRY(st.out(error, B, &p.error));
  }
  if (B > 0) {
    SMPC_HIP_CHECK(hipEventRecord(h->ev0, h->stream));
    SMPC_TRY(st.flush_up(h->stream));
    hipLaunchKernelGGL(smpc::smpc_plan_window_kernel, dim3((unsigned)B), dim3(smpc::kWave), 0, h->stream, p);
    SMPC_HIP_CHECK(hipGetLastError());
    SMPC_HIP_CHECK(hipEventRecord(h->ev1, h->stream));
    h->timed = true;
  }
  if (!in->on_device) {
    SMPC_TRY(st.down(window, p.window, B * L * 2, h->stream));
    SMPC_TRY(st.down(window_len, p.window_len, B, h->stream));
    SMPC_TRY(st.down(in->plan_start, static_cast<const int32_t*>(p.plan_start), B, h->stream));
    SMPC_TRY(st.down(error, p.error, B, h->stream));
    SMPC_TRY(st.finish(h->stream));
  }
  return SMPC_OK;
}

int smpc_select_command_batch(smpc_handle* h, int32_t B_, int32_t T, int32_t traj_rows, int32_t on_device, const int32_t* traj_n_poses,
                              const double* traj_cmds, const int32_t* status, const double* cmds, double* cmd_vel, int32_t* source,
                              const int32_t* window_error) {
  if (!h || !traj_cmds || !status || !cmds || !cmd_vel) { set_error("null handle / array"); return SMPC_ERR_INVALID_ARG; }
  if (B_ < 0 || T < 1 || traj_rows < 1) { set_error("bad B / T / traj_rows"); return SMPC_ERR_INVALID_ARG; }
  SMPC_HIP_CHECK(hipSetDevice(h->device));
  const size_t B = B_;
  smpc::SelectParams p;
  std::memset(&p, 0, sizeof(p));
  p.B = B_; p.T = T; p.rows = traj_rows;
  Staging st(h);
  if (on_device) {
    p.traj_n = traj_n_poses; p.traj_cmds = traj_cmds; p.status = status; p.cmds = cmds; p.cmd_vel = cmd_vel; p.source = source;
    p.window_error = window_error;
  } else {
    SMPC_TRY(st.up(window_error, B, &p.window_error, h->stream));
    SMPC_TRY(st.up(traj_n_poses, B, &p.traj_n, h->stream));
    SMPC_TRY(st.up(traj_cmds, B * (size_t)traj_rows * 2, &p.traj_cmds, h->stream));
    SMPC_TRY(st.up(status, B, &p.status, h->stream));
    SMPC_TRY(st.up(cmds, B * ((size_t)T + 1) * 2, &p.cmds, h->stream));
    SMPC_TRY(st.out(cmd_vel, B * 2, &p.cmd_vel));
    SMPC_TRY(st.out(source, B, &p.source));
  }
  if (B > 0) {
    SMPC_HIP_CHECK(hipEventRecord(h->ev0, h->stream));
    SMPC_TRY(st.flush_up(h->stream));
    hipLaunchKernelGGL(smpc::smpc_select_command_kernel, dim3((unsigned)((B + 255) / 256)), dim3(256), 0, h->stream, p);
    SMPC_HIP_CHECK(hipGetLastError());
    SMPC_HIP_CHECK(hipEventRecord(h->ev1, h->stream));
    h->timed = true;
  }
  if (!on_device) {
    SMPC_TRY(st.down(cmd_vel, p.cmd_vel, B * 2, h->stream));
    SMPC_TRY(st.down(source, p.source, B, h->stream));
    SMPC_TRY(st.finish(h->stream));
  }
  return SMPC_OK;
}

int smpc_stage_people_batch(smpc_handle* h, const smpc_scene_batch* sb, double* records, double* aux) {
  if (!h || !sb || !records || !aux) { set_error("null handle / scene batch / output"); return SMPC_ERR_INVALID_ARG; }
  if (sb->B < 0 || sb->T < 1 || sb->N < 1) { set_error("bad B/T/N"); return SMPC_ERR_INVALID_ARG; }
  if (sb->T + 1 > smpc::kWave || sb->N > smpc::kWave) { set_error("T + 1 > 64 or N > 64 is not supported"); return SMPC_ERR_UNSUPPORTED; }
  if (!sb->pose0 || !sb->people) { set_error("null input array"); return SMPC_ERR_INVALID_ARG; }
  SMPC_HIP_CHECK(hipSetDevice(h->device));
  smpc::KParams k;
  std::memset(&k, 0, sizeof(k));
  k.B = sb->B; k.T = sb->T; k.N = sb->N;
  smpc::fill_math_table(&k.mt);
  const size_t B = sb->B, T = sb->T, N = sb->N;
  const size_t nrec = B * N * T * 4, naux = B * T * 2;
  Staging st(h);
  double *drec = records, *daux = aux;
  if (sb->on_device) {
    k.pose0 = sb->pose0; k.people = sb->people; k.has_people = sb->has_people;
  } else {
    SMPC_TRY(st.up(sb->pose0, B * 3, &k.pose0, h->stream));
    SMPC_TRY(st.up(sb->people, B * (T + 1) * 6 * N, &k.people, h->stream));
    SMPC_TRY(st.up(sb->has_people, B, &k.has_people, h->stream));
    SMPC_TRY(st.out(records, nrec, &drec));
    SMPC_TRY(st.out(aux, naux, &daux));
    SMPC_HIP_CHECK(hipMemsetAsync(drec, 0, nrec * sizeof(double), h->stream));  // scenes without people: defined bytes
    SMPC_HIP_CHECK(hipMemsetAsync(daux, 0, naux * sizeof(double), h->stream));
  }
  SMPC_HIP_CHECK(hipEventRecord(h->ev0, h->stream));
  SMPC_TRY(st.flush_up(h->stream));
  SMPC_TRY(launch_stage(h, k, drec, daux));
  SMPC_HIP_CHECK(hipEventRecord(h->ev1, h->stream));
  h->timed = true;
  if (!sb->on_device) {
    SMPC_TRY(st.down(records, drec, nrec, h->stream));
    SMPC_TRY(st.down(aux, daux, naux, h->stream));
    SMPC_TRY(st.finish(h->stream));
  }
  return SMPC_OK;
}

double smpc_fp64_peak_probe(smpc_handle* h, int32_t iters) {
  if (!h || iters < 1) { set_error("bad arguments to smpc_fp64_peak_probe"); return -1.0; }
  if (hipSetDevice(h->device) != hipSuccess) return -1.0;
  const int blocks = h->num_cu * 8, threads = 256;   // 8 waves per SIMD
  double* out = nullptr;
  if (hipMalloc(reinterpret_cast<void**>(&out), (size_t)blocks * threads * sizeof(double)) != hipSuccess) return -1.0;
  double best = -1.0;
  for (int rep = 0; rep < 3; ++rep) {
    (void)hipEventRecord(h->ev0, h->stream);
    hipLaunchKernelGGL(smpc::smpc_fp64_peak_kernel, dim3(blocks), dim3(threads), 0, h->stream, out, iters);
    (void)hipEventRecord(h->ev1, h->stream);
    if (hipEventSynchronize(h->ev1) != hipSuccess) break;
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, h->ev0, h->ev1) != hipSuccess || ms <= 0.f) break;
    const double flop = 2.0 * 64.0 * (double)iters * (double)blocks * (double)threads;
    const double tf = flop / (ms * 1e-3) / 1e12;
    if (tf > best) best = tf;
  }
  (void)hipFree(out);
  return best;
}

int smpc_math_probe(smpc_handle* h, int32_t fn, int32_t n, const double* a, const double* b, double* out0, double* out1) {
  if (!h || !a || !out0 || n < 0 || fn < 0 || fn > 8) { set_error("bad arguments to smpc_math_probe"); return SMPC_ERR_INVALID_ARG; }
  if ((fn == 1 || fn == 4 || fn == 8) && !b) { set_error("second argument array is null"); return SMPC_ERR_INVALID_ARG; }
  if (fn == 2 && !out1) { set_error("out1 is null for sincos"); return SMPC_ERR_INVALID_ARG; }
  SMPC_HIP_CHECK(hipSetDevice(h->device));
  smpc::ProbeParams p;
  std::memset(&p, 0, sizeof(p));
  p.fn = fn; p.n = n;
  smpc::fill_math_table(&p.mt);
  smpc::fill_atan_nodes(&p.an);
  Staging st(h);
  SMPC_TRY(st.up(a, (size_t)n * (fn == 7 ? 8 : 1), &p.a, h->stream));
  SMPC_TRY(st.up(b, (size_t)n, &p.b, h->stream));
  SMPC_TRY(st.out(out0, (size_t)n, &p.o0));
  SMPC_TRY(st.out(out1, (size_t)n, &p.o1));
  if (n > 0) {
    SMPC_TRY(st.flush_up(h->stream));
    hipLaunchKernelGGL(smpc::smpc_math_probe_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, p);
    SMPC_HIP_CHECK(hipGetLastError());
  }
  SMPC_TRY(st.down(out0, p.o0, (size_t)n, h->stream));
  SMPC_TRY(st.down(out1, p.o1, (size_t)n, h->stream));
  SMPC_TRY(st.finish(h->stream));
  return SMPC_OK;
}

int smpc_eval_batch(smpc_handle* h, const smpc_scene_batch* sb, const double* params, smpc_eval_batch_out* out) {
  Dims d;
  SMPC_TRY(validate(h, sb, &d));
  if (!out || !params) { set_error("null params / output"); return SMPC_ERR_INVALID_ARG; }
  if (out->row_order != 0 && out->row_order != 1) { set_error("row_order must be 0 or 1"); return SMPC_ERR_INVALID_ARG; }
  SMPC_HIP_CHECK(hipSetDevice(h->device));
  smpc::KParams k;
  fill_kparams(h, sb, d, &k);
  Staging st(h);
  SMPC_TRY(bind_inputs(h, sb, d, &k, &st));
  SMPC_TRY(bind_people(h, sb, k, &st));
  const size_t B = sb->B;
  if (sb->on_device) {
    k.e_x = params;
    k.e_residuals = out->residuals; k.e_jacobian = out->jacobian; k.e_cost = out->cost; k.e_gradient = out->gradient;
    k.e_row_order = out->row_order;
    return launch(h, true, k);
  }
  k.e_row_order = out->row_order;
  SMPC_TRY(st.up(params, B * d.P, &k.e_x, h->stream));
  SMPC_TRY(st.out(out->residuals, B * d.M, &k.e_residuals));
  SMPC_TRY(st.out(out->jacobian, B * d.M * d.P, &k.e_jacobian));
  SMPC_TRY(st.out(out->cost, B, &k.e_cost));
  SMPC_TRY(st.out(out->gradient, B * d.P, &k.e_gradient));
  SMPC_TRY(st.flush_up(h->stream));
  SMPC_TRY(launch(h, true, k));
  SMPC_TRY(st.down(out->residuals, k.e_residuals, B * d.M, h->stream));
  SMPC_TRY(st.down(out->jacobian, k.e_jacobian, B * d.M * d.P, h->stream));
  SMPC_TRY(st.down(out->cost, k.e_cost, B, h->stream));
  SMPC_TRY(st.down(out->gradient, k.e_gradient, B * d.P, h->stream));
  SMPC_TRY(st.finish(h->stream));
  return SMPC_OK;
}

}  // extern "C"
