"""Development probe: one case of tests/test_gpu_parity.py::test_randomised_parameter_sets in detail."""
import sys
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from test_gpu_parity import _random_params, cmd_err
from nav2_social_mpc_controller_amd.scenes import make_scenes
from nav2_social_mpc_controller_amd.solver import BatchSolver
from oracle import oracle_py as O
case = int(sys.argv[1])
rng = np.random.default_rng(7000 + case)
prm = _random_params(rng)
N = int(rng.integers(1, 12))
sc = make_scenes(prm, 48, N, seed=8000 + case, map_cells=int(rng.choice([60, 120, 200])), n_valid=int(rng.integers(1, N + 1)))
print(prm)
print("dims", prm.dims(sc.T), "N", N, "T", sc.T)
s = BatchSolver(prm)
ev_o, ev_g = O.evaluate(prm, sc, sc.init_params), s.evaluate(sc, sc.init_params)
print("J rel err", np.max(np.abs(ev_o["jacobian"] - ev_g["jacobian"]) / np.maximum(1.0, np.abs(ev_o["jacobian"]))),
      "r rel err", np.max(np.abs(ev_o["residuals"] - ev_g["residuals"]) / np.maximum(1.0, np.abs(ev_o["residuals"]))))
rz = O.solve(prm, sc, nthreads=16, theta_zero_convention=True)
rg = s.solve(sc)
e = cmd_err(rg["cmds"], rz["cmds"])
for i in np.argsort(e)[-5:]:
    print(f"scene {i}: dcmd {e[i]:.3e} marginal {rz['marginal_decisions'][i]} iters gpu {rg['iterations'][i]} oracle {rz['iterations'][i]} "
          f"evals gpu {rg['evaluations'][i]} status {rg['status'][i]}/{rz['status'][i]} reason {rg['reason'][i]} "
          f"cost gpu {rg['final_cost'][i]:.12e} oracle {rz['final_cost'][i]:.12e} rel {abs(rg['final_cost'][i]-rz['final_cost'][i])/rz['final_cost'][i]:.2e}")
    print("   params gpu", rg["params"][i], "\n   params ora", rz["params"][i])
