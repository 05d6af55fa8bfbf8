"""Development probe: the K1 rows of one tools/gpu_soak.py case against the oracle and the independent Python
restatement: the worst Jacobian entry, the scale of its row, and which of the two CPU statements the device is nearer to.
usage: python tools/gpu_jac_case.py <case> [--wide]"""
import sys
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import torch  # noqa: F401
from nav2_social_mpc_controller_amd.scenes import make_scenes
from nav2_social_mpc_controller_amd.solver import BatchSolver
from oracle import oracle_py as O
WIDE = "--wide" in sys.argv
args = [a for a in sys.argv[1:] if not a.startswith("--")]
sys.argv = sys.argv[:1]
import gpu_soak_lib as L  # noqa: E402
case = int(args[0])
rng = np.random.default_rng(7000 + case)
prm = L.wide_params(rng) if WIDE else L.random_params(rng)
N = int(rng.integers(1, 41 if WIDE else 12))
sc = make_scenes(prm, 48, N, seed=8000 + case, map_cells=int(rng.choice([60, 120, 200])), n_valid=int(rng.integers(1, N + 1)))
ev_o, ev_g = O.evaluate(prm, sc, sc.init_params), BatchSolver(prm).evaluate(sc, sc.init_params)
Jo, Jg = ev_o["jacobian"], ev_g["jacobian"]
err = np.abs(Jo - Jg) / np.maximum(1.0, np.abs(Jo))
print("dims", prm.dims(sc.T), "N", N, "max entry error", err.max())
order = np.argsort(-err.reshape(-1))[:6]
for flat in order:
    b, i, q = np.unravel_index(flat, err.shape)
    row = Jo[b, i]
    print(f"scene {b} row {i} col {q}: oracle {Jo[b, i, q]:.17g} device {Jg[b, i, q]:.17g} diff {Jg[b, i, q] - Jo[b, i, q]:.3e} "
          f"| row max |J| {np.abs(row).max():.3e} | residual oracle {ev_o['residuals'][b, i]:.6e} device {ev_g['residuals'][b, i]:.6e}")
try:
    from oracle import pyref
    b = int(np.unravel_index(order[0], err.shape)[0])
    rp = pyref.evaluate(prm, sc, b, sc.init_params[b])
    Jp = np.asarray(rp[1] if isinstance(rp, tuple) else rp["jacobian"])
    print("scene", b, ": max |device - pyref| / max(1, |pyref|)", (np.abs(Jg[b] - Jp) / np.maximum(1.0, np.abs(Jp))).max(),
          " max |oracle - pyref| / max(1, |pyref|)", (np.abs(Jo[b] - Jp) / np.maximum(1.0, np.abs(Jp))).max())
except Exception as ex:  # the Python restatement's interface is not this probe's business
    print("pyref comparison skipped:", repr(ex)[:200])
