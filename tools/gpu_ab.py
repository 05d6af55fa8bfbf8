"""Development probe: A/B two builds of libsmpc_hip.so in ONE process-per-variant run, alternating, fixed-40 mode (steadiest)."""
import os, subprocess, sys
variants = sys.argv[1:] or ["", "_base"]
code = r'''
import sys, numpy as np, torch
sys.path.insert(0, ".")
from nav2_social_mpc_controller_amd.params import OptimizerParams
from nav2_social_mpc_controller_amd.scenes import make_scenes
from nav2_social_mpc_controller_amd.solver import BatchSolver
p = OptimizerParams.readme()
sc = make_scenes(p, 8192, 8)
out = []
for prm in (p, p.replace(fixed_iterations=1)):
    s = BatchSolver(prm); sb, t = sc.to_device(); rb, rt = s.alloc_results(8192, sc.T)
    ms = []
    for i in range(6):
        s.solve_device(sb, rb); ms.append(s.last_kernel_ms())
    out.append(min(ms[1:]))
print("%.3f %.3f" % tuple(out))
'''
res = {v: [] for v in variants}
for rep in range(3):
    for v in variants:
        env = dict(os.environ, SMPC_LIB_PATH=os.path.join(os.getcwd(), f"nav2_social_mpc_controller_amd/csrc/libsmpc_hip{v}.so"))
        o = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True).stdout.strip().splitlines()[-1]
        res[v].append(tuple(float(x) for x in o.split()))
for v in variants:
    a = res[v]
    print(f"variant '{v}': solve min {min(x[0] for x in a):.3f} ms (runs {[x[0] for x in a]}), fixed40 min {min(x[1] for x in a):.3f} ms (runs {[x[1] for x in a]})")
