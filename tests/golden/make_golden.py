"""Generates the committed golden fixtures under tests/golden/ (run from the repo root, CPU only):

    python tests/golden/make_golden.py

Nothing here reads /root/reference: the reference has no fixtures to take ("parity unpinned").
Each fixture holds seeded synthetic scenes plus the outputs of BOTH independent CPU restatements:
  * oracle/pyref.py  (torch.float64 autograd + numpy LM): residuals / Jacobian at the initial parameters of
    every scene, and a full solve of the first `n_pyref_solves` scenes (slow: ~1 min per solve);
  * oracle/smpc_oracle.cpp (C++ dual numbers + LM): full solves of every scene.
tests/ check the oracle against the pyref numbers, and the HIP path against the oracle numbers.
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from nav2_social_mpc_controller_amd.params import OptimizerParams  # noqa: E402
from nav2_social_mpc_controller_amd.scenes import make_scenes  # noqa: E402
from oracle import oracle_py as O  # noqa: E402
from oracle import pyref  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))

readme = OptimizerParams.readme()
CASES = {
    # name: (params, make_scenes kwargs, n_pyref_solves)
    "ref_n3_phantom": (readme, dict(B=4, N=3, n_valid=2, map_cells=80, seed=11), 2),
    "cfg3_n8": (readme, dict(B=4, N=8, map_cells=80, seed=12), 1),
    "params_yaml_n3": (OptimizerParams.params_yaml(), dict(B=2, N=3, map_cells=80, seed=13), 1),
    "cfg1_nopeople_qr": (OptimizerParams.params_yaml().replace(control_horizon=18, linear_solver_type="DENSE_QR"),
                         dict(B=2, N=3, map_cells=80, seed=14, people_present=False), 1),
    "quirk_unbounded_last_block": (readme.replace(time_step=0.1), dict(B=2, N=3, map_cells=80, seed=15), 1),
    # the reference's shipped benchmark parameter sets (params/*_in_benchmark.yaml:104-149): 80 x 80 local costmap, three
    # agents (people_to_status pads to 3); obst_only has people present with social_weight = agent_angle_weight = 0
    "soc_work_obst_benchmark": (OptimizerParams.soc_work_obst_benchmark(),
                                dict(B=4, N=3, n_valid=2, map_cells=OptimizerParams.BENCHMARK_COSTMAP_CELLS, seed=16), 1),
    "obst_only_benchmark": (OptimizerParams.obst_only_benchmark(),
                            dict(B=4, N=3, map_cells=OptimizerParams.BENCHMARK_COSTMAP_CELLS, seed=17), 1),
}


def params_to_npz(p):
    from dataclasses import asdict
    return {f"prm_{k}": np.array(v) for k, v in asdict(p).items()}


def main():
    only = sys.argv[1:]
    for name, (p, kw, n_solve) in CASES.items():
        if only and name not in only:
            continue
        t0 = time.time()
        sc = make_scenes(p, **kw)
        CH, bl, nb, P, M, nbnd = p.dims(sc.T, True)
        out = {}
        # pyref: r, J at init for every scene (rows beyond a no-people scene's M stay zero)
        R = np.zeros((sc.B, M))
        J = np.zeros((sc.B, M, P))
        for b in range(sc.B):
            r, jj = pyref.evaluate(p, sc, b, sc.init_params[b])
            R[b, :r.shape[0]] = r
            J[b, :r.shape[0]] = jj
        out["pyref_residuals"], out["pyref_jacobian"] = R, J
        xs, costs, its, reasons = [], [], [], []
        for b in range(n_solve):
            s = pyref.solve(p, sc, b)
            xs.append(s["x"]); costs.append(s["cost"]); its.append(s["iterations"]); reasons.append(s["reason"])
        out["pyref_x"] = np.array(xs); out["pyref_cost"] = np.array(costs); out["pyref_iterations"] = np.array(its)
        out["pyref_reason"] = np.array(reasons)
        # C++ oracle: full solves
        ro = O.solve(p, sc)
        for k, v in ro.items():
            out[f"oracle_{k}"] = v
        # same oracle with theta := 0 for exactly equal velocities (the HIP path's convention; identical to the
        # literal run wherever oracle_sign_noise_events == 0)
        rz = O.solve(p, sc, theta_zero_convention=True)
        for k, v in rz.items():
            out[f"oraclez_{k}"] = v
        eo = O.evaluate(p, sc, sc.init_params)
        out["oracle_residuals"], out["oracle_jacobian"] = eo["residuals"], eo["jacobian"]
        scene_path = os.path.join(HERE, f"{name}_scenes.npz")
        sc.save(scene_path)
        np.savez_compressed(os.path.join(HERE, f"{name}_expected.npz"), **out, **params_to_npz(p))
        print(f"{name}: T={sc.T} P={P} M={M} pyref-vs-oracle max|dr|={np.abs(R - eo['residuals']).max():.2e} "
              f"max|dJ|={np.abs(J - eo['jacobian']).max():.2e} "
              f"max|dx|={np.abs(out['pyref_x'] - ro['params'][:n_solve]).max():.2e} "
              f"iters pyref {its} oracle {ro['iterations'][:n_solve].tolist()}  [{time.time() - t0:.0f}s]", flush=True)


if __name__ == "__main__":
    main()
