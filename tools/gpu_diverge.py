"""Development probe: which scenes diverge between HIP and the oracle, vs the oracle's sign-noise diagnostic."""
import sys
import numpy as np
sys.path.insert(0, ".")
from nav2_social_mpc_controller_amd.params import OptimizerParams
from nav2_social_mpc_controller_amd.scenes import make_scenes
from nav2_social_mpc_controller_amd.solver import BatchSolver
from oracle import oracle_py as O
p = OptimizerParams.readme()
for N, B in ((8, 1024), (4, 1024), (3, 512)):
    sc = make_scenes(p, B, N, seed=77)
    s = BatchSolver(p)
    ro = O.solve(p, sc, nthreads=16); rg = s.solve(sc)
    dc = np.abs(ro["cmds"] - rg["cmds"]).reshape(B, -1).max(axis=1)
    bad = dc > 1e-5
    noisy = ro["sign_noise_events"] > 0
    print(f"N={N}: diverged {bad.sum()}/{B}; oracle-flagged noisy {noisy.sum()}; diverged&noisy {(bad&noisy).sum()}; diverged&clean {(bad&~noisy).sum()}; clean max|dcmd| {dc[~noisy].max():.3e}")
    if (bad&~noisy).any():
        idx = np.where(bad&~noisy)[0][:5]
        print("   clean-but-diverged scenes", idx, dc[idx], "iters", ro["iterations"][idx], rg["iterations"][idx])
