"""Development probe: distribution of the wavefronts of a grid over XCDs / CUs / SIMDs (tools/native/dispatch_probe.hip).
usage: python tools/gpu_dispatch.py [blocks=2048] [threads=64]"""
import ctypes, collections, os, sys
import numpy as np
import torch  # noqa: F401  (one HIP runtime per process)
here = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(os.path.join(here, "native", "libdispatch_probe.so"))
blocks = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
threads = int(sys.argv[2]) if len(sys.argv) > 2 else 64
waves = blocks * (threads // 64)
out = np.zeros((waves, 4), np.uint32)
lib.dispatch_probe.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_longlong, ctypes.c_void_p]
for rep in range(2):
    rc = lib.dispatch_probe(blocks, threads, 2_000_000, out.ctypes.data)  # ~20 us at 100 MHz counter ticks
    assert rc == 0, rc
hw, xcc = out[:, 0], out[:, 1] & 0xF
# HW_ID (gfx9): wave_id [3:0], simd_id [5:4], pipe [7:6], cu_id [11:8], sh_id [12], se_id [15:13] (gfx940: [14:13] + more)
simd, cu, sh, se = (hw >> 4) & 3, (hw >> 8) & 0xF, (hw >> 12) & 1, (hw >> 13) & 7
per_simd = collections.Counter(zip(xcc.tolist(), se.tolist(), sh.tolist(), cu.tolist(), simd.tolist()))
per_cu = collections.Counter(zip(xcc.tolist(), se.tolist(), sh.tolist(), cu.tolist()))
per_xcc = collections.Counter(xcc.tolist())
print(f"{waves} waves ({blocks} x {threads}): XCDs used {len(per_xcc)} {sorted(per_xcc.items())}")
print(f"CUs used {len(per_cu)}; waves per CU histogram {sorted(collections.Counter(per_cu.values()).items())}")
print(f"SIMDs used {len(per_simd)}; waves per SIMD histogram {sorted(collections.Counter(per_simd.values()).items())}")
