"""Generates tests/golden/next_rows.npz: small input / expected-output vectors for the rows of SURVEY §8(f) — people
projection (f1), people_to_status + format_to_optimize + memory store (f2), trajectorize (f3) — from the plain numpy /
Python statements under oracle/ (run from the repo root, CPU only):

    python tests/golden/make_golden_next_rows.py

Nothing here reads /root/reference: the reference holds no fixtures for these steps (parity unpinned). The vectors pin the
checkers themselves (tests/test_next_rows_golden.py fails if a statement drifts) and give the GPU tests expectations
that do not depend on re-running the checkers."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle import pyref_format, pyref_sfm, pyref_trajectorize  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def f1_case(seed, N, n_valid, T=28, grid=120):
    from test_projection import make_case
    return make_case(seed, N=N, n_valid=n_valid, T=T, grid=grid)


def main():
    out = {}
    # ---- f1: project_people, theta := 0 convention for exactly equal velocities (the HIP kernel's), 4 scenes
    cases = [f1_case(201, 3, 2), f1_case(202, 3, 3), f1_case(203, 3, 0), f1_case(204, 3, 1)]
    out["f1_init"] = np.stack([c["init"] for c in cases])
    out["f1_path"] = np.stack([c["path"] for c in cases])
    out["f1_idx"] = np.stack([c["idx"] for c in cases])
    out["f1_origin"] = np.stack([c["origin"] for c in cases])
    out["f1_res"], out["f1_max_time"], out["f1_dt"] = np.float64(cases[0]["res"]), np.float64(1.5), np.float64(0.05)
    proj = []
    for c in cases:
        od = dict(width=c["grid"], height=c["grid"], resolution=c["res"], origin_x=c["origin"][0], origin_y=c["origin"][1], indexes=c["idx"])
        proj.append(pyref_sfm.project_people(c["init"], c["path"], od, c["max_time"], c["dt"], theta_zero_convention=True))
    out["f1_expected"] = np.stack(proj)  # [4][T+1][N][6]
    # ---- f2: people_to_status, format_to_optimize (two consecutive calls on one memory), memory store
    from test_format import make_inputs
    rng = np.random.default_rng(301)
    people = rng.normal(size=(6, 5, 5))
    count = np.array([0, 1, 2, 3, 4, 5], np.int32)
    out["f2_people"], out["f2_count"] = people, count
    out["f2_status"], out["f2_has_people"] = pyref_format.people_to_status(people, count, 3)
    B, T, nb = 6, 28, 3
    path, cmds, speed = make_inputs(302, B, T)
    path2, cmds2, speed2 = make_inputs(303, B, T)
    mem = pyref_format.new_memory(B, T)
    o1 = pyref_format.format_to_optimize(path, cmds, speed, mem, 1.0, 0.5, 0.05, nb)
    res_path, res_cmds, _ = make_inputs(304, B, T)
    status = np.array([0, 1, 2, 0, 2, 1], np.int32)
    pyref_format.memory_store(status, res_path, res_cmds, mem)
    o2 = pyref_format.format_to_optimize(path2, cmds2, speed2, mem, 0.7, 0.3, 0.05, nb)
    for k, v in dict(path=path, cmds=cmds, speed=speed, path2=path2, cmds2=cmds2, speed2=speed2, res_path=res_path,
                     res_cmds=res_cmds, res_status=status).items():
        out["f2_" + k] = v
    for k, v in o1.items():
        out["f2_call1_" + k] = v
    for k, v in o2.items():
        out["f2_call2_" + k] = v
    for k, v in mem.items():
        out["f2_memory_end_" + k] = v
    # ---- f3: trajectorize, non-omnidirectional and omnidirectional
    from test_trajectorize import make_plans
    plan, plan_len, pose = make_plans(401, 8, L=120)
    out["f3_plan"], out["f3_plan_len"], out["f3_pose"] = plan, plan_len, pose
    S1 = 31
    for omni in (0, 1):
        P = np.zeros((8, S1, 3)); Cm = np.zeros((8, S1, 3)); n = np.zeros(8, np.int32)
        for s in range(8):
            p, c, err = pyref_trajectorize.trajectorize(plan[s, :plan_len[s]], pose[s], bool(omni), 0.6, 0.4, 1.0, 0.05, 1.5)
            assert err == 0
            n[s] = p.shape[0]
            P[s, :n[s]] = p
            Cm[s, :n[s] - 1] = c
        out[f"f3_omni{omni}_path"], out[f"f3_omni{omni}_cmds"], out[f"f3_omni{omni}_n_poses"] = P, Cm, n
    np.savez_compressed(os.path.join(HERE, "next_rows.npz"), **out)
    print("wrote next_rows.npz:", {k: v.shape for k, v in out.items() if hasattr(v, "shape")})


if __name__ == "__main__":
    main()
