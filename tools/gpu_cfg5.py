"""Development probe: BASELINE configs[4] shape (N=16, T=38, CH=30, bl=6 -> P=10) on one GPU."""
import sys
import numpy as np, torch
sys.path.insert(0, ".")
from nav2_social_mpc_controller_amd.params import OptimizerParams
from nav2_social_mpc_controller_amd.scenes import make_scenes
from nav2_social_mpc_controller_amd.solver import BatchSolver
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
prm = OptimizerParams.readme().replace(control_horizon=30, max_time=2.0)
sc = make_scenes(prm, B, 16)
CH, bl, nb, P, M, _ = prm.dims(sc.T, True)
s = BatchSolver(prm)
sb, tens = sc.to_device()
rb, rt = s.alloc_results(B, sc.T)
eo, et = s.alloc_eval(B, sc.T)
keep = s.stage_people_device(sb)
for _ in range(3):
    s.eval_device(sb, tens["init_params"].data_ptr(), eo)
k1 = s.last_kernel_ms()
for _ in range(3):
    s.solve_device(sb, rb)
ms = s.last_kernel_ms()
ev = rt["evaluations"].cpu().numpy()
print(f"cfg5 B={B} T={sc.T} P={P} M={M}: K1 {k1*1e3:.1f} us ({k1*1e6/B:.1f} ns/sweep); solve {ms:.3f} ms -> {B/ms*1e3:.0f} solves/s, sweeps mean {ev.mean():.1f}, {ms*1e6/ev.sum():.1f} ns per scene-sweep")
