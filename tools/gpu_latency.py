"""Development probe: latency of the plugin-style call (B = 1, host pointers) and of small batches."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from nav2_social_mpc_controller_amd.params import OptimizerParams
from nav2_social_mpc_controller_amd.scenes import make_scenes
from nav2_social_mpc_controller_amd.solver import BatchSolver

prm = OptimizerParams.readme()
s = BatchSolver(prm)
for B, N in ((1, 3), (1, 8), (16, 8), (256, 8)):
    sc = make_scenes(prm, B, N, seed=77, map_cells=120)
    s.solve(sc)
    wall, kern = [], []
    for _ in range(20):
        t0 = time.perf_counter()
        out = s.solve(sc)
        wall.append(time.perf_counter() - t0)
        kern.append(s.last_kernel_ms())
    print(f"B={B} N={N}: host-pointer call median {np.median(wall)*1e3:.3f} ms (kernel {np.median(kern):.3f} ms), "
          f"sweeps {out['evaluations'].mean():.1f}, iterations {out['iterations'].mean():.1f}")
