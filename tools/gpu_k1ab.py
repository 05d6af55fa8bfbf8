"""Development probe: stand-alone K1 (staged people input) for several builds csrc/libsmpc_hip<suffix>.so, alternating,
with all outputs / without the Jacobian / without any row output (what the stores cost)."""
import os, subprocess, sys
variants = sys.argv[1:] or [""]
code = r'''
import sys, numpy as np, torch
sys.path.insert(0, ".")
from nav2_social_mpc_controller_amd.params import OptimizerParams
from nav2_social_mpc_controller_amd.scenes import make_scenes
from nav2_social_mpc_controller_amd.solver import BatchSolver
p = OptimizerParams.readme()
sc = make_scenes(p, 8192, 8)
s = BatchSolver(p); sb, t = sc.to_device(); eo, et = s.alloc_eval(8192, sc.T)
keep = s.stage_people_device(sb)
res = []
for mode in range(4):
    if mode == 1: eo.row_order = 1
    if mode == 2: eo.jacobian = None
    if mode == 3: eo.residuals = None
    ms = []
    for i in range(10):
        s.eval_device(sb, t["init_params"].data_ptr(), eo); ms.append(s.last_kernel_ms())
    res.append(float(np.median(ms[2:])) * 1e3)
rb, rt = s.alloc_results(8192, sc.T)
ms = []
for i in range(5):
    s.solve_device(sb, rb); ms.append(s.last_kernel_ms())
print("%.1f %.1f %.1f %.1f %.3f" % (res[0], res[1], res[2], res[3], min(ms[1:])))
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
    print(f"variant '{v}': K1 reference order / critic-major / no J / no rows (us), solve (ms, staged input): {res[v]}")
