"""Development probe: A/B of builds csrc/libsmpc_hip<suffix>.so over the single-GPU shapes of the bench, lone launches,
alternating variants: cfg3 (8192 x 8, T = 28), the same with 40 iterations forced, BASELINE configs[4] (8192 x 16, T = 38,
P = 10), the reference's params.yaml shape (N = 3, T = 38, P = 10), configs[1] (1024 x 4).
usage: python tools/gpu_abshapes.py "" _base [reps]"""
import os, subprocess, sys
args = sys.argv[1:]
reps = 2
if args and args[-1].isdigit():
    reps = int(args.pop())
variants = args or ["", "_base"]
code = r'''
import sys, numpy as np, torch
sys.path.insert(0, ".")
from nav2_social_mpc_controller_amd.params import OptimizerParams
from nav2_social_mpc_controller_amd.scenes import make_scenes
from nav2_social_mpc_controller_amd.solver import BatchSolver
p = OptimizerParams.readme()
out = []
for prm, B, N in ((p, 8192, 8), (p.replace(fixed_iterations=1), 8192, 8),
                  (p.replace(control_horizon=30, max_time=2.0), 8192, 16), (OptimizerParams.params_yaml(), 8192, 3),
                  (p, 1024, 4)):
    sc = make_scenes(prm, B, N)
    s = BatchSolver(prm); sb, t = sc.to_device(); rb, rt = s.alloc_results(B, sc.T)
    ms = []
    for i in range(5):
        s.solve_device(sb, rb); ms.append(s.last_kernel_ms())
    out.append(min(ms[1:]))
    out.append(float(rt["evaluations"].sum().item()))
print(" ".join("%.4f" % v for v in out))
'''
names = ["cfg3", "fixed40", "cfg5", "params_yaml", "cfg2_1024"]
res = {v: [] for v in variants}
for rep in range(reps):
    for v in variants:
        env = dict(os.environ, SMPC_LIB_PATH=os.path.join(os.getcwd(), f"nav2_social_mpc_controller_amd/csrc/libsmpc_hip{v}.so"))
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
        o = r.stdout.strip().splitlines()
        if not o:
            print(v, "FAILED", r.stderr[-800:], flush=True); continue
        res[v].append([float(x) for x in o[-1].split()])
        print(f"rep {rep} variant '{v}':", o[-1], flush=True)
for v in variants:
    a = res[v]
    if not a:
        continue
    line = ", ".join(f"{n} {min(x[2 * i] for x in a):.3f} ms ({int(a[0][2 * i + 1])} sweeps)" for i, n in enumerate(names))
    print(f"variant '{v}': {line}")
