"""Development probe: plan-window kernel (PathHandler::transformGlobalPlan for B robots) alone, device-resident inputs."""
import sys
import numpy as np, torch
sys.path.insert(0, ".")
from nav2_social_mpc_controller_amd.episode import arc_plans
from nav2_social_mpc_controller_amd.params import OptimizerParams
from nav2_social_mpc_controller_amd.solver import BatchSolver
from nav2_social_mpc_controller_amd._abi import SmpcPlanWindowBatch
B = 8192
L = int(sys.argv[1]) if len(sys.argv) > 1 else 400
rng = np.random.default_rng(0)
pose = np.stack([rng.uniform(-5, 5, B), rng.uniform(-5, 5, B), rng.uniform(-3, 3, B)], 1)
plan, plan_len = arc_plans(pose, rng.uniform(-0.1, 0.1, B), L=L)
s = BatchSolver(OptimizerParams.readme())
dev = "cuda:0"
t = {"plan": torch.from_numpy(plan).to(dev), "len": torch.from_numpy(plan_len).to(dev), "pose": torch.from_numpy(pose).to(dev),
     "start": torch.zeros(B, dtype=torch.int32, device=dev), "win": torch.zeros((B, L, 2), dtype=torch.float64, device=dev),
     "wlen": torch.zeros(B, dtype=torch.int32, device=dev), "err": torch.zeros(B, dtype=torch.int32, device=dev)}
wb = SmpcPlanWindowBatch()
wb.B, wb.L, wb.on_device = B, L, 1
wb.max_robot_pose_search_dist, wb.dist_threshold = 5.0, 5.0
wb.plan, wb.plan_len, wb.plan_start, wb.robot_pose = t["plan"].data_ptr(), t["len"].data_ptr(), t["start"].data_ptr(), t["pose"].data_ptr()
ms = []
for _ in range(6):
    s.transform_global_plan_device(wb, t["win"].data_ptr(), t["wlen"].data_ptr(), t["err"].data_ptr())
    ms.append(s.last_kernel_ms())
print(f"plan window B={B} L={L}: ms", [round(m, 3) for m in ms], "window mean", t["wlen"].double().mean().item(), "errors", int((t["err"] != 0).sum()))
