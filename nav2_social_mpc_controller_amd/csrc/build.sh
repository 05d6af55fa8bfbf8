#!/bin/bash
# Builds libsmpc_hip.so (HIP kernels + C ABI) for gfx950, in-tree.
set -euo pipefail
cd "$(dirname "$0")"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -o libsmpc_hip.so smpc_hip.hip "$@"
