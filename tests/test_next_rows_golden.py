"""Committed vectors for the SURVEY §8(f) rows (tests/golden/next_rows.npz, made by tests/golden/make_golden_next_rows.py).
CPU: the numpy / Python statements and the C++ host mirrors reproduce them (a drifting checker fails here). GPU: the
kernels behind the C ABI reproduce them without re-running any checker."""
import os

import numpy as np
import pytest

from conftest import GOLDEN


@pytest.fixture(scope="module")
def G():
    return dict(np.load(os.path.join(GOLDEN, "next_rows.npz"), allow_pickle=False))


def yaw_close(a, b, tol):
    d = np.abs(a - b)
    return np.max(np.minimum(d, np.abs(d - 2 * np.pi))) <= tol


# ------------------------------------------------------------------------------------------------ CPU: checkers
def test_pyref_statements_reproduce_the_vectors(G):
    from oracle import pyref_format, pyref_sfm, pyref_trajectorize
    for s in range(G["f1_init"].shape[0]):
        grid = G["f1_idx"].shape[-1]
        od = dict(width=grid, height=grid, resolution=float(G["f1_res"]), origin_x=G["f1_origin"][s, 0],
                  origin_y=G["f1_origin"][s, 1], indexes=G["f1_idx"][s])
        got = pyref_sfm.project_people(G["f1_init"][s], G["f1_path"][s], od, float(G["f1_max_time"]), float(G["f1_dt"]),
                                       theta_zero_convention=True)
        assert np.max(np.abs(got - G["f1_expected"][s])) <= 1e-12
    st, has = pyref_format.people_to_status(G["f2_people"], G["f2_count"], 3)
    assert np.array_equal(st, G["f2_status"]) and np.array_equal(has, G["f2_has_people"])
    B, Tp, _ = G["f2_path"].shape
    mem = pyref_format.new_memory(B, Tp - 1)
    o1 = pyref_format.format_to_optimize(G["f2_path"], G["f2_cmds"], G["f2_speed"], mem, 1.0, 0.5, 0.05, 3)
    pyref_format.memory_store(G["f2_res_status"], G["f2_res_path"], G["f2_res_cmds"], mem)
    o2 = pyref_format.format_to_optimize(G["f2_path2"], G["f2_cmds2"], G["f2_speed2"], mem, 0.7, 0.3, 0.05, 3)
    assert (o1.pop("T_scene") == Tp - 1).all() and (o2.pop("T_scene") == Tp - 1).all()   # fixed horizon (no n_poses)
    for k in o1:
        assert np.max(np.abs(o1[k] - G["f2_call1_" + k])) <= 1e-14 and np.max(np.abs(o2[k] - G["f2_call2_" + k])) <= 1e-14
    for k in mem:
        assert np.array_equal(mem[k], G["f2_memory_end_" + k])
    for omni in (0, 1):
        for s in range(G["f3_plan"].shape[0]):
            p, c, err = pyref_trajectorize.trajectorize(G["f3_plan"][s, :G["f3_plan_len"][s]], G["f3_pose"][s], bool(omni),
                                                        0.6, 0.4, 1.0, 0.05, 1.5)
            n = G[f"f3_omni{omni}_n_poses"][s]
            assert err == 0 and p.shape[0] == n
            assert np.max(np.abs(p - G[f"f3_omni{omni}_path"][s, :n])) <= 1e-13
            assert np.max(np.abs(c - G[f"f3_omni{omni}_cmds"][s, :n - 1])) <= 1e-13


# ------------------------------------------------------------------------------------------------ GPU: the C ABI
@pytest.fixture(scope="module")
def solver():
    from nav2_social_mpc_controller_amd.params import OptimizerParams
    from nav2_social_mpc_controller_amd.solver import BatchSolver
    return BatchSolver(OptimizerParams.readme())


@pytest.mark.gpu
def test_gpu_project_people_reproduces_the_vectors(G, solver):
    got, err = solver.project_people(G["f1_init"], G["f1_path"], G["f1_idx"], G["f1_origin"], float(G["f1_res"]),
                                     float(G["f1_max_time"]), float(G["f1_dt"]))
    assert np.all(err == 0)
    assert np.max(np.abs(got.transpose(0, 1, 3, 2) - G["f1_expected"])) <= 1e-9   # [B][T+1][6][N] -> [B][T+1][N][6]


@pytest.mark.gpu
def test_gpu_format_chain_reproduces_the_vectors(G, solver):
    st, has = solver.people_to_status(G["f2_people"], G["f2_count"], 3)
    assert np.max(np.abs(st - G["f2_status"])) <= 1e-14 and np.array_equal(has, G["f2_has_people"])
    B, Tp, _ = G["f2_path"].shape
    mem = solver.new_memory(B, Tp - 1)
    o1 = solver.format_to_optimize(G["f2_path"], G["f2_cmds"], G["f2_speed"], mem, 1.0, 0.5)
    solver.memory_store(G["f2_res_status"], G["f2_res_path"], G["f2_res_cmds"], mem)
    o2 = solver.format_to_optimize(G["f2_path2"], G["f2_cmds2"], G["f2_speed2"], mem, 0.7, 0.3)
    for o, tag in ((o1, "f2_call1_"), (o2, "f2_call2_")):
        assert (o.pop("T_scene") == Tp - 1).all()
        for k in o:
            assert yaw_close(o[k], G[tag + k], 1e-13), (tag, k)
    for k in mem:
        assert np.array_equal(mem[k], G["f2_memory_end_" + k]), k


@pytest.mark.gpu
@pytest.mark.parametrize("omni", [0, 1])
def test_gpu_trajectorize_reproduces_the_vectors(G, solver, omni):
    from nav2_social_mpc_controller_amd.params import TrajectorizerParams
    tp = TrajectorizerParams(omnidirectional=bool(omni), desired_linear_vel=0.6, lookahead_dist=0.4, max_angular_vel=1.0,
                             time_step=0.05, max_time=1.5)
    got = solver.trajectorize(tp, G["f3_plan"], G["f3_plan_len"], G["f3_pose"])
    want_p, want_c, n = G[f"f3_omni{omni}_path"], G[f"f3_omni{omni}_cmds"], G[f"f3_omni{omni}_n_poses"]
    assert np.array_equal(got["n_poses"], n) and np.all(got["error"] == 0)
    assert np.max(np.abs(got["path"][:, :, :2] - want_p[:, :, :2])) <= 1e-11 and yaw_close(got["path"][:, :, 2], want_p[:, :, 2], 1e-11)
    assert np.max(np.abs(got["cmds"] - want_c[:, :, [0, 2]])) <= 1e-11 and np.max(np.abs(got["cmds_vy"] - want_c[:, :, 1])) <= 1e-11
