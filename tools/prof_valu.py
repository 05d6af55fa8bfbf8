"""Profiling target: VALU instruction counts of solve launches on workloads that isolate parts of a trip
(run under rocprofv3 --pmc SQ_INSTS_VALU ...; dispatches appear in this order)."""
import sys
import torch
sys.path.insert(0, ".")
from nav2_social_mpc_controller_amd.params import OptimizerParams
from nav2_social_mpc_controller_amd.scenes import make_scenes
from nav2_social_mpc_controller_amd.solver import BatchSolver
p = OptimizerParams.readme()
B = 8192
for name, prm, N, kw in (("cfg3", p, 8, {}), ("nopeople", p, 3, {"people_present": False}), ("fixed40", p.replace(fixed_iterations=1), 8, {}),
                         ("n1", p, 1, {})):
    sc = make_scenes(prm, B, N, **kw)
    s = BatchSolver(prm)
    sb, tens = sc.to_device()
    keep = s.stage_people_device(sb) if sc.has_people.any() else None
    rb, rt = s.alloc_results(B, sc.T)
    s.solve_device(sb, rb)
    torch.cuda.synchronize()
    ev = int(rt["evaluations"].sum().item()); it = int(rt["iterations"].sum().item())
    print(f"{name}: sweeps {ev} iterations {it} kernel ms {s.last_kernel_ms():.3f}", flush=True)
