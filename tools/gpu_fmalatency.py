"""Development probe: how often can one wavefront issue a dependent v_fma_f64? (tools/native/latency_probe.hip)
Prints ns per FMA instruction of one wave for 1 / 2 / 4 / 8 independent chains, with 1 wave per SIMD (1024 blocks),
2 and 4 waves per SIMD: the gap between 1 chain and 8 chains is what instruction-level parallelism can buy a kernel
that runs at low occupancy."""
import ctypes, os, time
import numpy as np
import torch  # noqa: F401
here = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(os.path.join(here, "native", "liblatency_probe.so"))
lib.latency_probe.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
iters = 2000
for blocks in (1, 1024, 2048, 4096):
    for chains in (1, 2, 4, 8):
        cyc = np.zeros((blocks, 2), np.int64)
        rc = lib.latency_probe(chains, blocks, iters, cyc.ctypes.data)
        assert rc == 0, rc
        n = iters * 16 * chains
        ticks = cyc[:, 0].mean()  # 100 MHz ticks
        print(f"blocks {blocks:5d} ({blocks/1024:.0f} waves/SIMD) chains {chains}: {ticks * 10.0 / n:.3f} ns per FMA instruction of a wave "
              f"= {ticks * 10.0 / n * 2.4:.1f} cycles at 2.4 GHz; clock64 ratio {cyc[:,1].mean()/max(cyc[:,0].mean(),1):.2f}")
