"""Profiling target: two solves of one of the one-scene-per-wave shapes (cfg5: N = 16, T = 38, P = 10; yaml: the
reference's params.yaml, N = 3), device-resident. usage: prof_target_shapes.py cfg5|yaml [B]"""
import sys
import torch
sys.path.insert(0, ".")
from nav2_social_mpc_controller_amd.params import OptimizerParams
from nav2_social_mpc_controller_amd.scenes import make_scenes
from nav2_social_mpc_controller_amd.solver import BatchSolver
which = sys.argv[1] if len(sys.argv) > 1 else "cfg5"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
prm, N = (OptimizerParams.readme().replace(control_horizon=30, max_time=2.0), 16) if which == "cfg5" else (OptimizerParams.params_yaml(), 3)
sc = make_scenes(prm, B, N)
s = BatchSolver(prm)
sb, tens = sc.to_device()
rb, rt = s.alloc_results(B, sc.T)
for _ in range(2):
    s.solve_device(sb, rb)
torch.cuda.synchronize()
print(which, "evals", int(rt["evaluations"].sum().item()), "kernel ms", s.last_kernel_ms())
