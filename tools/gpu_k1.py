"""Development probe: stand-alone K1 (residual + Jacobian sweep) timing at a few batch sizes."""
import sys
import numpy as np, torch
sys.path.insert(0, ".")
from nav2_social_mpc_controller_amd.params import OptimizerParams
from nav2_social_mpc_controller_amd.scenes import make_scenes
from nav2_social_mpc_controller_amd.solver import BatchSolver

p = OptimizerParams.readme()
for B, N in ((8192, 8), (32768, 8), (8192, 16), (8192, 0)):
    sc = make_scenes(p, B, max(N, 1), people_present=N > 0)
    CH, bl, nb, P, M, _ = p.dims(sc.T, True)
    s = BatchSolver(p)
    sb, tens = sc.to_device()
    eo, et = s.alloc_eval(B, sc.T)
    if N > 0:
        keep = s.stage_people_device(sb)
    torch.cuda.synchronize()
    ms = []
    for i in range(12):
        s.eval_device(sb, tens["init_params"].data_ptr(), eo)
        ms.append(s.last_kernel_ms())
    NN = sc.N
    by = 8 * (6 * NN * sc.T + 2 * (sc.T + 1) + P + 5) + 16 * sc.T + 8 * (M * P + M)
    m = float(np.median(ms[2:]))
    print(f"K1 B={B} N={N}: median {m*1e3:.1f} us min {min(ms)*1e3:.1f} us -> {B*by/m/1e6:.0f} GB/s algorithmic")
