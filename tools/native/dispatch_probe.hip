// Development probe (not part of the library): where do the wavefronts of a grid of small workgroups land?
// Every wavefront records its hardware id (XCC, SE, CU, SIMD) and spins for a while so that the whole grid is
// resident together; the host script counts wavefronts per SIMD.
// build: hipcc --offload-arch=gfx950 -O2 -fPIC -shared -o tools/native/libdispatch_probe.so tools/native/dispatch_probe.hip
#include <hip/hip_runtime.h>
#include <stdint.h>

__global__ void probe_kernel(uint32_t* out, long long spin, int vgpr_pad) {
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) / 64;
  uint32_t hw, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  const long long t0 = __builtin_readcyclecounter();
  double a = threadIdx.x;
  while (__builtin_readcyclecounter() - t0 < spin) a = a * 1.0000001 + 1e-9;
  if ((threadIdx.x & 63) == 0) {
    out[4 * wave] = hw; out[4 * wave + 1] = xcc; out[4 * wave + 2] = (uint32_t)(t0 & 0xffffffff); out[4 * wave + 3] = (uint32_t)a;
  }
}

extern "C" int dispatch_probe(int blocks, int threads, long long spin, uint32_t* host_out) {
  uint32_t* d;
  const size_t n = (size_t)blocks * (threads / 64) * 4;
  if (hipMalloc(&d, n * sizeof(uint32_t)) != hipSuccess) return 1;
  hipLaunchKernelGGL(probe_kernel, dim3(blocks), dim3(threads), 0, 0, d, spin, 0);
  if (hipDeviceSynchronize() != hipSuccess) return 2;
  hipMemcpy(host_out, d, n * sizeof(uint32_t), hipMemcpyDeviceToHost);
  hipFree(d);
  return 0;
}
