"""Development probe: throughput of back-to-back solves when consecutive batches alternate between 2..3 streams
(the tail of one persistent solve kernel overlaps the head of the next)."""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from nav2_social_mpc_controller_amd.params import OptimizerParams
from nav2_social_mpc_controller_amd.scenes import make_scenes
from nav2_social_mpc_controller_amd.solver import BatchSolver
p = OptimizerParams.readme()
B = 8192
sc = make_scenes(p, B, 8)
sb, tens = sc.to_device()
for nstreams in (1, 2, 3, 4):
    solvers = [BatchSolver(p) for _ in range(nstreams)]
    streams = [torch.cuda.Stream() for _ in range(nstreams)]
    outs = []
    for s, st in zip(solvers, streams):
        s.set_stream(st.cuda_stream)
        outs.append(s.alloc_results(B, sc.T))
    for i in range(nstreams):
        solvers[i].solve_device(sb, outs[i][0])
    torch.cuda.synchronize()
    K = 12
    t0 = time.perf_counter()
    for k in range(K):
        solvers[k % nstreams].solve_device(sb, outs[k % nstreams][0])
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ref = outs[0][1]["cmds"].cpu().numpy()
    same = all(np.array_equal(ref, o[1]["cmds"].cpu().numpy()) for o in outs)
    print(f"streams {nstreams}: {dt/K*1e3:.3f} ms per batch -> {B*K/dt:.0f} solves/s; identical outputs {same}")
