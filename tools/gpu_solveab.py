"""Development probe: A/B of builds csrc/libsmpc_hip<suffix>.so on the solve kernel: a lone launch and the 4-stream
overlapped rate of the bench (raw people input, i.e. with the library's staging pass), alternating variants."""
import os, subprocess, sys
variants = sys.argv[1:] or [""]
code = r'''
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
import os
if os.environ.get("PROBED_STREAMS") == "1":
    from nav2_social_mpc_controller_amd.episode import concurrent_streams
    streams = concurrent_streams(n, "cuda:0")
else:
    streams = [torch.cuda.Stream() for _ in range(n)]
outs = []
for s, st in zip(solvers, streams):
    s.set_stream(st.cuda_stream); outs.append(s.alloc_results(B, sc.T))
    if os.environ.get("SHARE"): s.set_solve_share(int(os.environ["SHARE"]))
for i in range(n): solvers[i].solve_device(sb, outs[i][0])
torch.cuda.synchronize()
K = 24
t0 = time.perf_counter()
for k in range(K): solvers[k % n].solve_device(sb, outs[k % n][0])
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / K
lone = []
for i in range(4):
    solvers[0].solve_device(sb, outs[0][0]); torch.cuda.synchronize(); lone.append(solvers[0].last_kernel_ms())
print("%.3f %.3f" % (dt * 1e3, min(lone)))
'''
res = {v: [] for v in variants}
for rep in range(2):
    for v in variants:
        env = dict(os.environ, SMPC_LIB_PATH=os.path.join(os.getcwd(), f"nav2_social_mpc_controller_amd/csrc/libsmpc_hip{v}.so"))
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
        o = r.stdout.strip().splitlines()
        if not o:
            print(v, "FAILED", r.stderr[-600:]); continue
        res[v].append(tuple(float(x) for x in o[-1].split()))
for v in variants:
    print(f"variant '{v}': (ms per 8192-scene batch overlapped on 4 streams, lone launch ms): {res[v]}")
