"""Development probe: which scenes diverge between HIP and the oracle, and does it correlate with standing agents?"""
import sys
import numpy as np
sys.path.insert(0, ".")
from nav2_social_mpc_controller_amd.params import OptimizerParams
from nav2_social_mpc_controller_amd.scenes import make_scenes
from nav2_social_mpc_controller_amd.solver import BatchSolver
from oracle import oracle_py as O
p = OptimizerParams.readme()
for sf in (0.2, 0.0):
    sc = make_scenes(p, 512, 8, standing_fraction=sf)
    s = BatchSolver(p)
    ro = O.solve(p, sc, nthreads=16); rg = s.solve(sc)
    dc = np.abs(ro["cmds"] - rg["cmds"]).reshape(512, -1).max(axis=1)
    bad = dc > 1e-5
    vzero = (ro["params"][:, 0::2] == 0.0).any(axis=1)
    print(f"standing_fraction={sf}: diverged {bad.sum()}/512; of which oracle solution has a v==0 block: {(bad & vzero).sum()}; scenes with v==0 overall {vzero.sum()}")
    print("   iteration mismatch among diverged:", (ro["iterations"][bad] != rg["iterations"][bad]).sum(), " final cost rel diff (diverged) median", np.median(np.abs(ro["final_cost"][bad]-rg["final_cost"][bad])/ro["final_cost"][bad]) if bad.any() else 0)
    print("   worst scenes:", np.argsort(-dc)[:6], dc[np.argsort(-dc)[:6]])
