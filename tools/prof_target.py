"""Profiling target: a few K1 sweeps and fused solves at BASELINE config 3 (B=8192, N=8), device-resident."""
import sys
import torch
sys.path.insert(0, ".")
from nav2_social_mpc_controller_amd.params import OptimizerParams
from nav2_social_mpc_controller_amd.scenes import make_scenes
from nav2_social_mpc_controller_amd.solver import BatchSolver
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
N = int(sys.argv[2]) if len(sys.argv) > 2 else 8
nsolve = int(sys.argv[3]) if len(sys.argv) > 3 else 2
p = OptimizerParams.readme()
sc = make_scenes(p, B, N)
s = BatchSolver(p)
sb, tens = sc.to_device()
rb, rt = s.alloc_results(B, sc.T)
eo, et = s.alloc_eval(B, sc.T, row_order=1)
keep = s.stage_people_device(sb)   # K1 and the solve kernel read the staged people block
for _ in range(3):
    s.eval_device(sb, tens["init_params"].data_ptr(), eo)
for _ in range(nsolve):
    s.solve_device(sb, rb)
torch.cuda.synchronize()
print("evals", int(rt["evaluations"].sum().item()), "kernel ms", s.last_kernel_ms())
