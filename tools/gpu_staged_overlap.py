"""Development probe: does the staging pass cost the 4-stream rate anything? The overlapped loop of the bench with the
reference-layout people block (staging kernel inside every step) and with a pre-staged block (solve kernel only)."""
import sys, time, numpy as np, torch
sys.path.insert(0, ".")
from nav2_social_mpc_controller_amd.params import OptimizerParams
from nav2_social_mpc_controller_amd.scenes import make_scenes
from nav2_social_mpc_controller_amd.solver import BatchSolver
p = OptimizerParams.readme()
B = 8192
sc = make_scenes(p, B, 8)
sb, tens = sc.to_device()
n = 4
solvers = [BatchSolver(p) for _ in range(n)]
streams = [torch.cuda.Stream() for _ in range(n)]
outs = []
for s, st in zip(solvers, streams):
    s.set_stream(st.cuda_stream); outs.append(s.alloc_results(B, sc.T))
keep = solvers[0].stage_people_device(sb)   # fills sb.people_records / people_aux
rec, aux = sb.people_records, sb.people_aux
torch.cuda.synchronize()
for label, (r, a) in (("staging pass inside every step", (None, None)), ("pre-staged people block", (rec, aux)), ("staging pass inside every step", (None, None)), ("pre-staged people block", (rec, aux))):
    sb.people_records, sb.people_aux = r, a
    for i in range(n): solvers[i].solve_device(sb, outs[i][0])
    torch.cuda.synchronize()
    K = 32
    t0 = time.perf_counter()
    for k in range(K): solvers[k % n].solve_device(sb, outs[k % n][0])
    torch.cuda.synchronize()
    print(f"{label}: {(time.perf_counter() - t0) / K * 1e3:.3f} ms per 8192-scene step on 4 streams")
