"""Development probe: one case of tools/gpu_soak.py in detail — the scenes over the tolerance, their iteration counts and
cost gaps, and how far the ORACLE moves on them under 1e-15 relative perturbations of the start pose (20 samples).
usage: python tools/gpu_soak_case.py <case> [--wide] [--horizons]"""
import sys
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import torch  # noqa: F401
from conftest import cmd_err
from nav2_social_mpc_controller_amd.scenes import make_scenes
from nav2_social_mpc_controller_amd.solver import BatchSolver
from oracle import oracle_py as O
WIDE, HORIZONS = "--wide" in sys.argv, "--horizons" in sys.argv
args = [a for a in sys.argv[1:] if not a.startswith("--")]
sys.argv = sys.argv[:1]
import gpu_soak_lib as L  # noqa: E402
case = int(args[0])
rng = np.random.default_rng(7000 + case)
prm = L.wide_params(rng) if WIDE else L.random_params(rng)
N = int(rng.integers(1, 41 if WIDE else 12))
sc = make_scenes(prm, 48, N, seed=8000 + case, map_cells=int(rng.choice([60, 120, 200])), n_valid=int(rng.integers(1, N + 1)))
if HORIZONS:
    sc = sc.with_horizons(rng.integers(1, sc.T + 1, size=sc.B).astype(np.int32))
rz = O.solve(prm, sc, nthreads=16, theta_zero_convention=True)
rg = BatchSolver(prm).solve(sc)
e = cmd_err(rg["cmds"], rz["cmds"])
over = np.where(e > 1e-5)[0]
print("dims", prm.dims(sc.T), "N", N, "scenes over 1e-5:", over.tolist())
g = np.random.default_rng(99)
moves = np.zeros((20, sc.B))
for k in range(20):
    sc2 = sc.select(np.arange(sc.B))
    sc2.pose0 = sc.pose0 * (1.0 + 1e-15 * g.standard_normal(sc.pose0.shape))
    moves[k] = cmd_err(O.solve(prm, sc2, nthreads=16, theta_zero_convention=True)["cmds"], rz["cmds"])
for b in over:
    print(f"scene {b}: |dcmd| {e[b]:.2e}, marginal decisions {rz['marginal_decisions'][b]}, iterations device {rg['iterations'][b]} oracle {rz['iterations'][b]}, "
          f"status {rg['status'][b]}/{rz['status'][b]}, rel cost gap {(rg['final_cost'][b] - rz['final_cost'][b]) / rz['final_cost'][b]:.2e}; "
          f"the oracle itself moves by max {moves[:, b].max():.2e} (median {np.median(moves[:, b]):.2e}) under 1e-15 input perturbations")
