"""Development soak of the chain kernels over many more seeds than the suite runs: people projection against the numpy
restatement (tests/test_projection.py's generator) and trajectorize against the plain-Python restatement
(tests/test_trajectorize.py's generator), one line per failing case.
usage: python tools/gpu_soak_chain.py [seeds=40]"""
import sys, time
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import torch  # noqa: F401
from test_projection import make_case, pyref_project
from test_trajectorize import make_plans, pyref, params
from nav2_social_mpc_controller_amd.params import OptimizerParams
from nav2_social_mpc_controller_amd.solver import BatchSolver
seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 40
s = BatchSolver(OptimizerParams.readme())
t0 = time.time()
bad = 0
worst_p = 0.0
n_p = 0
for N, n_valid in ((3, 2), (3, 3), (5, 4), (8, 8), (8, 5), (12, 9), (16, 16)):
    cases = [make_case(5000 + 31 * N + i, N=N, n_valid=n_valid) for i in range(seeds)]
    got, err = s.project_people(np.stack([c["init"] for c in cases]), np.stack([c["path"] for c in cases]),
                                np.stack([c["idx"] for c in cases]), np.stack([c["origin"] for c in cases]),
                                cases[0]["res"], cases[0]["max_time"], cases[0]["dt"])
    for b, c in enumerate(cases):
        want = pyref_project(c, convention=True)
        e = float(np.max(np.abs(got[b].transpose(0, 2, 1) - want)))
        worst_p = max(worst_p, e); n_p += 1
        if err[b] != 0 or not e < 1e-9:
            bad += 1
            print(f"projection N={N} n_valid={n_valid} seed {5000 + 31 * N + b}: err flag {err[b]} max diff {e:.2e}", flush=True)
print(f"projection: {n_p} cases, worst difference {worst_p:.2e}, {time.time() - t0:.0f} s", flush=True)
worst_t = 0.0
n_t = 0
for omni in (False, True):
    for L, max_time in ((60, 1.5), (160, 1.5), (390, 2.0), (500, 3.0), (1200, 2.0)):
        tp = params(omni, desired_linear_vel=0.6, max_time=max_time)
        for rep in range(max(1, seeds // 20)):
            B = 64
            plan, plan_len, pose = make_plans(9000 + 100 * L + rep, B, L)
            got = s.trajectorize(tp, plan, plan_len, pose)
            for b in range(B):
                p, c, err = pyref(plan[b, :plan_len[b]], pose[b], tp)
                n_t += 1
                ok = got["error"][b] == err and (err == 1 or got["n_poses"][b] == p.shape[0])
                e = 0.0
                if ok and err != 1:
                    n = p.shape[0]
                    dy = np.abs(got["path"][b, :n, 2] - p[:, 2])
                    e = max(np.max(np.abs(got["path"][b, :n, :2] - p[:, :2])), np.max(np.minimum(dy, np.abs(dy - 2 * np.pi))))
                    if n > 1:
                        e = max(e, np.max(np.abs(got["cmds"][b, :n - 1] - c[:, [0, 2]])), np.max(np.abs(got["cmds_vy"][b, :n - 1] - c[:, 1])))
                worst_t = max(worst_t, float(e))
                if not ok or not e <= 1e-11:
                    bad += 1
                    print(f"trajectorize omni={omni} L={L} max_time={max_time} seed {9000 + 100 * L + rep} plan {b}: ok={ok} diff {e:.2e}", flush=True)
print(f"trajectorize: {n_t} plans, worst difference {worst_t:.2e}, {time.time() - t0:.0f} s")
# plan window (PathHandler::transformGlobalPlan): tests/test_path_window.py's generator and checker over more seeds
from test_path_window import make_plans as make_window_plans, check as check_window
n_w = 0
worst_w = 0.0
for rep in range(max(1, seeds // 4)):
    for L in (30, 256, 700):
        B = 96
        plan, plan_len, pose = make_window_plans(31000 + 17 * rep + L, B, L)
        rng = np.random.default_rng(rep * 7 + L)
        search, thr = float(rng.uniform(0.5, 6.0)), float(rng.uniform(1.0, 8.0))
        to_local = None if rep % 2 else np.stack([rng.uniform(-2, 2, B), rng.uniform(-2, 2, B), rng.uniform(-3, 3, B)], 1)
        start = np.zeros(B, np.int32)
        for tick in range(3):
            before = start.copy()
            got = s.transform_global_plan(plan, plan_len, start, pose, search, thr, to_local)
            try:
                worst_w = max(worst_w, check_window(got, plan, plan_len, before, start, pose, search, thr, to_local))
            except AssertionError as e:
                bad += 1
                print(f"plan window rep {rep} L={L} tick {tick}: {e}", flush=True)
            n_w += B
            nxt = np.minimum(start + 5, np.maximum(plan_len - 1, 0))
            pose[:, :2] = plan[np.arange(B), nxt] + 0.04
print(f"plan window: {n_w} calls, worst difference {worst_w:.2e}, {time.time() - t0:.0f} s")
print(f"soak (chain): {bad} failing cases")
