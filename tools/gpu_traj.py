"""Development probe: trajectorize kernel timing."""
import sys
import numpy as np
sys.path.insert(0, ".")
from nav2_social_mpc_controller_amd.episode import arc_plans
from nav2_social_mpc_controller_amd.params import OptimizerParams, TrajectorizerParams
from nav2_social_mpc_controller_amd.solver import BatchSolver
from nav2_social_mpc_controller_amd._abi import SmpcTrajectorizeOut
import torch
B = 8192
rng = np.random.default_rng(0)
pose = np.stack([rng.uniform(-5, 5, B), rng.uniform(-5, 5, B), rng.uniform(-3, 3, B)], 1)
L = int(sys.argv[2]) if len(sys.argv) > 2 else 400
kmax = min(0.24, 3.0 / (L * 0.05))  # the arc stays short of a full circle
plan, plan_len = arc_plans(pose, rng.uniform(-kmax, kmax, B), L=L)
tp = TrajectorizerParams(desired_linear_vel=0.6, max_time=float(sys.argv[1]) if len(sys.argv) > 1 else 1.5)
s = BatchSolver(OptimizerParams.readme())
dev = "cuda:0"
tb = s.trajectorize_c(tp, B, L, 1)
t = {"plan": torch.from_numpy(plan).to(dev), "len": torch.from_numpy(plan_len).to(dev), "pose": torch.from_numpy(pose).to(dev)}
S1 = tp.max_steps + 1
f64 = dict(dtype=torch.float64, device=dev)
o = {"path": torch.zeros((B, S1, 3), **f64), "cmds": torch.zeros((B, S1, 2), **f64), "vy": torch.zeros((B, S1), **f64),
     "n": torch.zeros(B, dtype=torch.int32, device=dev), "e": torch.zeros(B, dtype=torch.int32, device=dev)}
to = SmpcTrajectorizeOut()
tb.plan, tb.plan_len, tb.robot_pose = t["plan"].data_ptr(), t["len"].data_ptr(), t["pose"].data_ptr()
to.path, to.cmds, to.cmds_vy, to.n_poses, to.error = o["path"].data_ptr(), o["cmds"].data_ptr(), o["vy"].data_ptr(), o["n"].data_ptr(), o["e"].data_ptr()
ms = []
for _ in range(6):
    s.trajectorize_device(tb, to)
    ms.append(s.last_kernel_ms())
print(f"trajectorize B=8192 L={L} max_steps={tp.max_steps}: ms", [round(m, 3) for m in ms], "n_poses mean", o["n"].double().mean().item(), "errors", int((o["e"] != 0).sum()))
