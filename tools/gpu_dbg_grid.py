import sys, os
sys.path.insert(0, ".")
os.environ["SMPC_DEBUG_GRID"] = "1"
from nav2_social_mpc_controller_amd.params import OptimizerParams
from nav2_social_mpc_controller_amd.scenes import make_scenes
from nav2_social_mpc_controller_amd.solver import BatchSolver
p = OptimizerParams.readme()
for prm, B, N in ((p, 8192, 8), (p.replace(control_horizon=30, max_time=2.0), 8192, 16), (OptimizerParams.params_yaml(), 8192, 3)):
    sc = make_scenes(prm, B, N)
    s = BatchSolver(prm); sb, t = sc.to_device(); rb, rt = s.alloc_results(B, sc.T)
    s.solve_device(sb, rb)
    print("shape", B, N, sc.T, "kernel ms", s.last_kernel_ms(), flush=True)
