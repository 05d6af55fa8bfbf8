"""Development probe: the NB = 5 / W = 64 shapes (cfg5 and the reference's params.yaml) on the current library
(set SMPC_LIB_PATH for a -DSMPC_STAMPS build to get the phase shares on stderr)."""
import sys
import numpy as np, torch
sys.path.insert(0, ".")
from nav2_social_mpc_controller_amd.params import OptimizerParams
from nav2_social_mpc_controller_amd.scenes import make_scenes
from nav2_social_mpc_controller_amd.solver import BatchSolver
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
for name, prm, N in (("cfg5", OptimizerParams.readme().replace(control_horizon=30, max_time=2.0), 16),
                     ("params_yaml", OptimizerParams.params_yaml(), 3)):
    sc = make_scenes(prm, B, N)
    s = BatchSolver(prm)
    sb, tens = sc.to_device()
    rb, rt = s.alloc_results(B, sc.T)
    for _ in range(3):
        s.solve_device(sb, rb)
    ms = s.last_kernel_ms()
    ev = rt["evaluations"].cpu().numpy(); it = rt["iterations"].cpu().numpy()
    print(f"{name}: solve {ms:.3f} ms -> {B/ms*1e3:.0f} solves/s, sweeps mean {ev.mean():.1f} iters {it.mean():.1f}, {ms*1e6/ev.sum():.1f} ns per scene-sweep", flush=True)
