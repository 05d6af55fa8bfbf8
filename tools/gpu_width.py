"""Development probe: a solve launch of a two-scenes-per-wave shape (T + 1 <= 32) run as one scene per wave with helper
lanes (W = 64) against the plain W = 32 kernel, by batch size: kernel time, and the largest difference of the commands.
usage (GPU box): python tools/gpu_width.py [N ...]"""
import os
import sys

import numpy as np

sys.path.insert(0, ".")
from nav2_social_mpc_controller_amd.params import OptimizerParams
from nav2_social_mpc_controller_amd.scenes import make_scenes
from nav2_social_mpc_controller_amd.solver import BatchSolver

prm = OptimizerParams.readme()
s = BatchSolver(prm)
crowd = [int(a) for a in sys.argv[1:]] or [8, 4]
for N in crowd:
    for B in (1, 16, 256, 1024, 2048, 3072, 4096):
        sc = make_scenes(prm, B, N, seed=77, map_cells=120)
        res = {}
        for width in ("32", "64"):
            os.environ["SMPC_SOLVE_WIDTH"] = width
            s.solve(sc)
            kern = []
            for _ in range(8):
                out = s.solve(sc)
                kern.append(s.last_kernel_ms())
            res[width] = (float(np.median(kern)), out)
        d = np.abs(res["32"][1]["cmds"] - res["64"][1]["cmds"]).reshape(B, -1).max(axis=1)
        same_it = (res["32"][1]["iterations"] == res["64"][1]["iterations"]).mean()
        print(f"N={N} B={B}: W=32 {res['32'][0]:.3f} ms, W=64 {res['64'][0]:.3f} ms ({res['64'][0] / res['32'][0]:.2f}x), "
              f"sweeps {res['32'][1]['evaluations'].mean():.1f}; |dcmd| median {np.median(d):.1e} max {d.max():.1e}, "
              f"equal iteration counts {100 * same_it:.1f} %", flush=True)
