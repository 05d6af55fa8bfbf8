"""Development probe: a lone launch of the headline batch under resident waves per CU x attained-service priority step
(SMPC_LONE_WAVES_PER_CU x SMPC_PRIO_STEP, both read by the library at every launch): does priority for the old waves
make a third wave per SIMD pay for a launch that has the GPU to itself?
usage: python tools/gpu_lone_sweep.py [shape: cfg3 | cfg5 | yaml]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, ".")
from nav2_social_mpc_controller_amd.params import OptimizerParams
from nav2_social_mpc_controller_amd.scenes import make_scenes
from nav2_social_mpc_controller_amd.solver import BatchSolver

shape = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
p = OptimizerParams.readme()
prm, N = {"cfg3": (p, 8), "cfg5": (p.replace(control_horizon=30, max_time=2.0), 16),
          "yaml": (OptimizerParams.params_yaml(), 3)}[shape]
B = 8192
sc = make_scenes(prm, B, N)
s = BatchSolver(prm)
sb, t = sc.to_device()
rb, rt = s.alloc_results(B, sc.T)
for _ in range(2):
    s.solve_device(sb, rb)
for waves in ("6", "8", "10", "12"):
    for prio in ("0", "8", "16", "32", "64"):
        os.environ["SMPC_LONE_WAVES_PER_CU"] = waves
        os.environ["SMPC_PRIO_STEP"] = prio
        ms = []
        for _ in range(5):
            s.solve_device(sb, rb)
            ms.append(s.last_kernel_ms())
        print(f"{shape}: waves per CU {waves:>2s} prio step {prio:>2s}: lone launch min {min(ms):.3f} ms  median {sorted(ms)[2]:.3f} ms", flush=True)
