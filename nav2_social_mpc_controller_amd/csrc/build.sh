#!/bin/bash
# Builds libsmpc_hip.so (HIP kernels + C ABI) for gfx950, in-tree.
set -euo pipefail
cd "$(dirname "$0")"
# -disable-machine-licm: inside the persistent solve loop machine-LICM hoists dozens of 64-bit literals (libm polynomial
# coefficients) and uniform kernel arguments out of the loop; they overflow the SGPR file, spill to VGPR lanes / scratch and
# come back as v_readlane + scratch loads inside the sweep. Without it the solve kernel has no VGPR spills (241 VGPRs) and
# runs 13% faster (measured, round 1).
# -target-feature -fmacf64-inst: without v_fmac_f64 every fused multiply-add is the three-address v_fma_f64, which takes a
# polynomial coefficient straight from an SGPR pair; with it the compiler puts each coefficient into the accumulator with
# two v_mov_b32 first (round 1: 53 % of the sweep's VALU issue was not FP64 arithmetic). The host pass ignores the flag.
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -mllvm -disable-machine-licm \
  -Xclang -target-feature -Xclang -fmacf64-inst -Wno-pass-failed -o "${SMPC_OUT:-libsmpc_hip.so}" smpc_hip.hip "$@"
