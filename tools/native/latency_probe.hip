// Development probe (not part of the library): issue interval of dependent FP64 FMAs on one wavefront.
// One wavefront per SIMD (grid = 1024 blocks of 64 threads would be 1 per SIMD; here a single block: the wave has its
// SIMD to itself), `chains` independent accumulator chains per lane, `iters` rounds; cycles from s_memtime.
// build: hipcc --offload-arch=gfx950 -O2 -fPIC -shared -o tools/native/liblatency_probe.so tools/native/latency_probe.hip
#include <hip/hip_runtime.h>
#include <stdint.h>

template <int C>
__global__ void fma_chain(double* out, long long* cyc, int iters, double a, double b) {
  double x[C];
#pragma unroll
  for (int c = 0; c < C; ++c) x[c] = threadIdx.x * 1e-3 + c;
  const long long t0 = __builtin_readcyclecounter();  // s_memtime: constant 100 MHz on gfx9; wall clock, not shader clock
  const long long c0 = clock64();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
#pragma unroll
      for (int c = 0; c < C; ++c) x[c] = __builtin_fma(x[c], a, b);
    }
  }
  const long long c1 = clock64();
  const long long t1 = __builtin_readcyclecounter();
  double s = 0;
#pragma unroll
  for (int c = 0; c < C; ++c) s += x[c];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) { cyc[2 * blockIdx.x] = t1 - t0; cyc[2 * blockIdx.x + 1] = c1 - c0; }
}

extern "C" int latency_probe(int chains, int blocks, int iters, long long* host_cyc) {
  double* d; long long* c;
  if (hipMalloc(&d, (size_t)blocks * 64 * sizeof(double)) != hipSuccess) return 1;
  if (hipMalloc(&c, (size_t)blocks * 2 * sizeof(long long)) != hipSuccess) return 1;
  for (int rep = 0; rep < 2; ++rep) {
    switch (chains) {
      case 1: hipLaunchKernelGGL(fma_chain<1>, dim3(blocks), dim3(64), 0, 0, d, c, iters, 0.999999, 1e-7); break;
      case 2: hipLaunchKernelGGL(fma_chain<2>, dim3(blocks), dim3(64), 0, 0, d, c, iters, 0.999999, 1e-7); break;
      case 4: hipLaunchKernelGGL(fma_chain<4>, dim3(blocks), dim3(64), 0, 0, d, c, iters, 0.999999, 1e-7); break;
      case 8: hipLaunchKernelGGL(fma_chain<8>, dim3(blocks), dim3(64), 0, 0, d, c, iters, 0.999999, 1e-7); break;
      default: return 3;
    }
    if (hipDeviceSynchronize() != hipSuccess) return 2;
  }
  hipMemcpy(host_cyc, c, (size_t)blocks * 2 * sizeof(long long), hipMemcpyDeviceToHost);
  hipFree(d); hipFree(c);
  return 0;
}
