"""Development probe: find the first LM iteration where HIP and oracle part ways for the worst scene of a case."""
import sys
import numpy as np
sys.path.insert(0, ".")
from nav2_social_mpc_controller_amd.params import OptimizerParams
from nav2_social_mpc_controller_amd.scenes import make_scenes
from nav2_social_mpc_controller_amd.solver import BatchSolver
from oracle import oracle_py as O
README = OptimizerParams.readme()
prm = README.replace(control_horizon=30, max_time=2.0)
sc = make_scenes(prm, 128, 16, seed=204)
rg = BatchSolver(prm).solve(sc)
rz = O.solve(prm, sc, nthreads=16, theta_zero_convention=True)
err = np.abs(rg["cmds"] - rz["cmds"]).reshape(128, -1).max(axis=1)
b = int(np.argmax(err))
print("worst scene", b, err[b], "iters gpu/oracle", rg["iterations"][b], rz["iterations"][b], "cost", rg["final_cost"][b], rz["final_cost"][b], "events", rz["sign_noise_events"][b])
one = sc.select([b])
O.set_theta_zero_convention(True)
tr = O.trace(prm, one, 0)
O.set_theta_zero_convention(False)
np.set_printoptions(linewidth=220, precision=10)
for k in range(0, 41):
    p2 = prm.replace(max_iterations=k)
    g = BatchSolver(p2).solve(one)
    o = O.solve(p2, one, theta_zero_convention=True)
    d = np.abs(g["params"] - o["params"]).max()
    print(f"cap {k:2d}: max|dparams| {d:.3e} cost gpu {g['final_cost'][0]:.12e} oracle {o['final_cost'][0]:.12e} iters {g['iterations'][0]} {o['iterations'][0]} reason {g['reason'][0]} {o['reason'][0]} evals {g['evaluations'][0]} {o['evaluations'][0]}")
    if d > 1e-6:
        break
print("oracle trace rows [iter, cost, cost_change, gmax, step_norm, rho, radius, ls_evals, accepted]:")
print(tr[max(0, k - 3):k + 2])
