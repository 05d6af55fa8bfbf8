"""Warm start / input formatting: Optimizer::format_to_optimize + TrajectoryMemory. Three statements
are compared: oracle/pyref_format.py (numpy, the checker), the C++ host adapter (CPU) and the HIP kernels behind
smpc_format_to_optimize_batch / smpc_memory_store_batch (GPU). PARITY UNPINNED: the reference holds no fixtures for it."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "nav2_social_mpc_controller_amd", "host")


def make_inputs(seed, B, T):
    rng = np.random.default_rng(seed)
    path = np.zeros((B, T + 1, 3))
    path[:, :, 0:2] = rng.uniform(-5, 5, size=(B, 1, 2)) + np.cumsum(rng.uniform(-0.03, 0.03, size=(B, T + 1, 2)), axis=1)
    path[:, :, 2] = rng.uniform(-np.pi, np.pi, size=(B, 1)) + np.cumsum(rng.uniform(-0.03, 0.03, size=(B, T + 1)), axis=1)
    cmds = np.stack([rng.uniform(0, 0.6, size=(B, T + 1)), rng.uniform(-1, 1, size=(B, T + 1))], axis=-1)
    speed = np.stack([rng.uniform(0, 0.6, size=B), rng.uniform(-1, 1, size=B)], axis=-1)
    return path, cmds, speed


@pytest.fixture(scope="module")
def hostlib():
    subprocess.check_call(["make", "-C", HOST, "-s"])
    lib = C.CDLL(os.path.join(HOST, "libsmpc_host.so"))
    lib.smpc_host_format_to_optimize.restype = C.c_int
    lib.smpc_host_format_to_optimize.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
                                                 C.c_float, C.c_float, C.c_float, C.c_float, C.c_void_p]
    return lib


def host_format(lib, path, cmds, prev_path, prev_cmds, speed, wp, wc, max_time, dt):
    n = path.shape[0]
    out = np.zeros((n, 6))
    nprev = 0 if prev_path is None else prev_path.shape[0]
    pp = np.zeros((1, 3)) if prev_path is None else np.ascontiguousarray(prev_path)
    pc = np.zeros((1, 2)) if prev_cmds is None else np.ascontiguousarray(prev_cmds)
    m = lib.smpc_host_format_to_optimize(np.ascontiguousarray(path).ctypes.data, np.ascontiguousarray(cmds).ctypes.data, n,
                                         pp.ctypes.data, pc.ctypes.data, nprev, np.ascontiguousarray(speed).ctypes.data,
                                         wp, wc, max_time, dt, out.ctypes.data)
    return out[:m]


@pytest.mark.parametrize("wp,wc", [(1.0, 0.5), (0.7, 0.3), (1.0, 1.0)])
def test_pyref_matches_host_adapter(hostlib, wp, wc):
    from oracle import pyref_format
    B, T, nb = 6, 28, 3
    path, cmds, speed = make_inputs(1, B, T)
    prev_path, prev_cmds, _ = make_inputs(2, B, T)
    mem = pyref_format.new_memory(B, T)
    mem["prev_path"][3:], mem["prev_cmds"][3:], mem["valid"][3:] = prev_path[3:], prev_cmds[3:], 1  # half the records filled
    mem0 = {k: v.copy() for k, v in mem.items()}
    out = pyref_format.format_to_optimize(path, cmds, speed, mem, wp, wc, 0.05, nb)
    for s in range(B):
        have = mem0["valid"][s] != 0
        ref = host_format(hostlib, path[s], cmds[s], mem0["prev_path"][s] if have else None,
                          mem0["prev_cmds"][s] if have else None, speed[s], wp, wc, 10.0, 0.05)
        assert ref.shape == (T + 1, 6)
        assert np.max(np.abs(ref - out["robot_status"][s])) <= 1e-15
    # an empty record is filled with the incoming path / cmds, a filled one is left alone
    assert np.array_equal(mem["prev_path"][:3], path[:3]) and np.array_equal(mem["prev_cmds"][:3], cmds[:3])
    assert np.array_equal(mem["prev_path"][3:], prev_path[3:]) and mem["valid"].tolist() == [1] * B
    assert np.array_equal(out["path_pts"], out["robot_status"][:, :, 0:2])
    assert np.array_equal(out["goal_yaw"], out["robot_status"][:, T, 2])
    assert np.array_equal(out["init_params"].reshape(B, nb, 2), out["robot_status"][:, :nb, 4:6])
    assert np.array_equal(out["robot_status"][:, 0, 4:6], speed)


def test_host_adapter_cut_matches_rollout_steps(hostlib):
    """format_to_optimize cuts a long path to round(max_time / dt) - 1 poses (src/optimizer.cpp:492-497);
    OptimizerParams.rollout_steps is that count minus the popped velocity."""
    from nav2_social_mpc_controller_amd.params import OptimizerParams
    for prm in (OptimizerParams.readme(), OptimizerParams.params_yaml()):
        n = 80
        path, cmds, speed = make_inputs(3, 1, n - 1)
        ref = host_format(hostlib, path[0], cmds[0], None, None, speed[0], 1.0, 1.0, prm.max_time, prm.time_step)
        assert ref.shape[0] - 1 == prm.rollout_steps
        short = host_format(hostlib, path[0][:10], cmds[0][:10], None, None, speed[0], 1.0, 1.0, prm.max_time, prm.time_step)
        assert short.shape[0] == 10


def test_pyref_memory_store():
    from oracle import pyref_format
    B, T = 5, 8
    path, cmds, _ = make_inputs(4, B, T)
    mem = pyref_format.new_memory(B, T)
    status = np.array([0, 1, 2, 0, 2], np.int32)
    pyref_format.memory_store(status, path, cmds, mem)
    assert mem["valid"].tolist() == [1, 1, 0, 1, 0]
    assert np.array_equal(mem["prev_path"][[0, 1, 3]], path[[0, 1, 3]]) and not mem["prev_path"][[2, 4]].any()


@pytest.mark.gpu
@pytest.mark.parametrize("wp,wc,T", [(1.0, 0.5, 28), (0.7, 0.3, 38), (1.0, 1.0, 5)])
def test_gpu_format_matches_pyref(wp, wc, T):
    from nav2_social_mpc_controller_amd.params import OptimizerParams
    from nav2_social_mpc_controller_amd.solver import BatchSolver
    from oracle import pyref_format
    prm = OptimizerParams.readme()
    s = BatchSolver(prm)
    B = 300
    CH, bl, nb, P, M, _ = prm.dims(T, True)
    path, cmds, speed = make_inputs(5, B, T)
    prev_path, prev_cmds, _ = make_inputs(6, B, T)
    mem = s.new_memory(B, T)
    mem["prev_path"][::2], mem["prev_cmds"][::2], mem["valid"][::2] = prev_path[::2], prev_cmds[::2], 1
    mem_ref = {k: v.copy() for k, v in mem.items()}
    got = s.format_to_optimize(path, cmds, speed, mem, wp, wc)
    exp = pyref_format.format_to_optimize(path, cmds, speed, mem_ref, wp, wc, prm.time_step, nb)
    for k in exp:
        err = np.abs(got[k] - exp[k])
        if k in ("robot_status", "pose0", "goal_yaw"):  # yaws: compare on the circle
            err = np.minimum(err, np.abs(err - 2 * np.pi))
        assert np.max(err) <= 1e-13, k
    for k in mem:
        assert np.array_equal(mem[k], mem_ref[k]), k
    # second call: every record is now valid and is blended against
    path2, cmds2, speed2 = make_inputs(7, B, T)
    got2 = s.format_to_optimize(path2, cmds2, speed2, mem, wp, wc)
    exp2 = pyref_format.format_to_optimize(path2, cmds2, speed2, mem_ref, wp, wc, prm.time_step, nb)
    assert np.max(np.abs(got2["robot_status"][:, :, [0, 1, 3, 4, 5]] - exp2["robot_status"][:, :, [0, 1, 3, 4, 5]])) <= 1e-13
    assert np.array_equal(mem["prev_path"], mem_ref["prev_path"])


@pytest.mark.gpu
def test_gpu_memory_store_matches_pyref():
    from nav2_social_mpc_controller_amd.params import OptimizerParams
    from nav2_social_mpc_controller_amd.solver import BatchSolver
    from oracle import pyref_format
    s = BatchSolver(OptimizerParams.readme())
    B, T = 257, 28
    path, cmds, _ = make_inputs(8, B, T)
    status = np.random.default_rng(9).integers(0, 3, size=B).astype(np.int32)
    mem = s.new_memory(B, T)
    mem_ref = {k: v.copy() for k, v in mem.items()}
    s.memory_store(status, path, cmds, mem)
    pyref_format.memory_store(status, path, cmds, mem_ref)
    for k in mem:
        assert np.array_equal(mem[k], mem_ref[k]), k
    assert (status == 2).any() and (mem["valid"] == (status != 2)).all()


def test_pyref_people_to_status_pads_and_truncates():
    from oracle import pyref_format
    people = np.random.default_rng(10).normal(size=(4, 5, 5))
    out, has = pyref_format.people_to_status(people, np.array([0, 2, 3, 5]), 3)
    assert has.tolist() == [0, 1, 1, 1]
    assert (out[0, :, 3] == -1).all() and out[1, :2, 3].tolist() == [0, 0] and out[1, 2, 3] == -1 and (out[3, :, 3] == 0).all()
    assert np.allclose(out[3, 2, 4], np.hypot(people[3, 2, 2], people[3, 2, 3]))


@pytest.mark.gpu
@pytest.mark.parametrize("N", [3, 8])
def test_gpu_people_to_status_matches_pyref(N):
    from nav2_social_mpc_controller_amd.params import OptimizerParams
    from nav2_social_mpc_controller_amd.solver import BatchSolver
    from oracle import pyref_format
    rng = np.random.default_rng(11)
    B, Np = 300, 10
    people = rng.normal(size=(B, Np, 5))
    people[::7, :, 2:4] = 0.0  # standing people: yaw = atan2(0, 0) = 0
    count = rng.integers(0, Np + 1, size=B).astype(np.int32)
    got, has = BatchSolver(OptimizerParams.readme()).people_to_status(people, count, N)
    exp, ehas = pyref_format.people_to_status(people, count, N)
    assert np.array_equal(has, ehas) and np.max(np.abs(got - exp)) <= 1e-14


def fov_case(seed, B=200, Np=8):
    rng = np.random.default_rng(seed)
    pose = np.stack([rng.uniform(3, 7, B), rng.uniform(3, 7, B), rng.uniform(-np.pi, np.pi, B)], 1)
    people = rng.normal(size=(B, Np, 5))
    r, phi = rng.uniform(0.5, 6.0, (B, Np)), rng.uniform(-np.pi, np.pi, (B, Np))
    people[:, :, 0] = pose[:, None, 0] + r * np.cos(phi)
    people[:, :, 1] = pose[:, None, 1] + r * np.sin(phi)
    count = rng.integers(0, Np + 1, size=B).astype(np.int32)
    origin = np.array([[0.0, 0.0]])
    return pose, people, count, origin, 200, 200, 0.05   # 10 m x 10 m costmap: some persons fall outside


def pyref_fov_status(pose, people, count, origin, sx, sy, res, fov, N):
    from oracle import pyref_format
    B = pose.shape[0]
    st, has = np.zeros((B, N, 6)), np.zeros(B, np.uint8)
    kept = []
    for s in range(B):
        keep = pyref_format.fov_filter(people[s], count[s], pose[s], fov, origin[0], sx, sy, res)
        kept.append(len(keep))
        sel = people[s][keep][None] if keep else np.zeros((1, 1, 5))
        o, h = pyref_format.people_to_status(sel, np.array([len(keep)]), N)
        st[s], has[s] = o[0], h[0]
    return st, has, np.array(kept)


def test_pyref_fov_filter_keeps_only_people_ahead_and_on_the_map():
    pose, people, count, origin, sx, sy, res = fov_case(21, B=50)
    st, has, kept = pyref_fov_status(pose, people, count, origin, sx, sy, res, np.pi / 4, 3)
    assert 0 < kept.sum() < count.sum() and (kept <= count).all()
    for s in range(50):
        for a in range(min(kept[s], 3)):
            bearing = np.arctan2(st[s, a, 1] - pose[s, 1], st[s, a, 0] - pose[s, 0])
            d = (bearing - pose[s, 2] + np.pi) % (2 * np.pi) - np.pi
            assert abs(d) < np.pi / 4 + 1e-6 and 0 <= st[s, a, 0] < 10 and 0 <= st[s, a, 1] < 10


@pytest.mark.gpu
@pytest.mark.parametrize("fov", [np.pi / 4, 1.3])
def test_gpu_fov_filter_matches_pyref(fov):
    from nav2_social_mpc_controller_amd.params import OptimizerParams
    from nav2_social_mpc_controller_amd.solver import BatchSolver
    pose, people, count, origin, sx, sy, res = fov_case(22)
    got, has = BatchSolver(OptimizerParams.readme()).people_to_status(people, count, 3, robot_pose=pose, fov_angle=fov,
                                                                      costmap_origin=origin, size_x=sx, size_y=sy, resolution=res)
    exp, ehas, kept = pyref_fov_status(pose, people, count, origin, sx, sy, res, fov, 3)
    assert np.array_equal(has, ehas) and np.max(np.abs(got - exp)) <= 1e-14
    assert (kept == 0).any() and (kept > 3).any()


@pytest.mark.gpu
def test_gpu_select_command_applies_the_reference_fallbacks():
    """computeVelocityCommands' returned command: optimised cmds[0], else the trajectorizer's first command
    (src/social_mpc_controller.cpp:241-245), else 0.1 m/s straight (:180-189); nothing when transformGlobalPlan threw
    (src/path_handler.cpp:44-47, 100-103). A path shorter than the batch's horizon is NOT a fallback case."""
    from nav2_social_mpc_controller_amd.params import OptimizerParams
    from nav2_social_mpc_controller_amd.solver import BatchSolver
    rng = np.random.default_rng(12)
    B, T, rows = 300, 28, 31
    traj_cmds, cmds = rng.normal(size=(B, rows, 2)), rng.normal(size=(B, T + 1, 2))
    status = rng.integers(0, 3, size=B).astype(np.int32)
    n = rng.choice([0, 5, T + 1, rows], size=B).astype(np.int32)
    s = BatchSolver(OptimizerParams.readme())
    got, src = s.select_command(n, traj_cmds, status, cmds)
    want_src = np.where(n <= 0, 2, np.where(status == 2, 1, 0))
    want = np.where((want_src == 2)[:, None], np.array([0.1, 0.0]), np.where((want_src == 1)[:, None], traj_cmds[:, 0], cmds[:, 0]))
    assert np.array_equal(src, want_src) and np.array_equal(got, want)
    assert set(want_src.tolist()) == {0, 1, 2} and ((n == 5) & (src == 0)).any()
    werr = rng.choice([0, 0, 0, 1, 2], size=B).astype(np.int32)
    got2, src2 = s.select_command(n, traj_cmds, status, cmds, window_error=werr)
    assert np.array_equal(src2, np.where(werr != 0, 3, want_src))
    assert np.array_equal(got2, np.where((werr != 0)[:, None], 0.0, want)) and (src2 == 3).any()


def ragged_inputs(seed, B, rows, max_poses):
    """paths of n poses each (n - 1 commands), n spread over 0, 1, 2 .. rows with the cut boundary well covered"""
    rng = np.random.default_rng(seed)
    path, cmds, speed = make_inputs(seed, B, rows - 1)
    n = rng.integers(2, rows + 1, size=B)
    n[:8] = [0, 1, 2, max_poses - 1, max_poses, max_poses + 1, rows, 3]
    return path, cmds, speed, n.astype(np.int32)


def test_pyref_with_path_lengths_matches_host_adapter(hostlib):
    """Horizons per scene: the numpy statement against the transliterated Optimizer::format_to_optimize, scene by scene,
    with incoming paths and memory records of every length (cut at round(max_time / dt), blending only while
    i < previous_path.poses.size())."""
    from oracle import pyref_format
    max_time, dt = 1.5, 0.05
    max_poses = int(np.round(np.float32(max_time) / np.float32(dt)))
    T, rows, nb, B = max_poses - 1, max_poses + 1, 3, 64
    path, cmds, speed, n = ragged_inputs(31, B, rows, max_poses)
    prev_path, prev_cmds, _ = make_inputs(32, B, T)
    rng = np.random.default_rng(33)
    mem = pyref_format.new_memory(B, T, lengths=True)
    plen = rng.integers(1, T + 2, size=B)
    for s in range(B):
        if s % 3:   # two thirds of the records hold a previous solution of some horizon (path and commands: same size)
            mem["prev_path"][s, :plen[s]], mem["prev_cmds"][s, :plen[s]] = prev_path[s, :plen[s]], prev_cmds[s, :plen[s]]
            mem["valid"][s], mem["length"][s] = 1, (plen[s], plen[s])
    mem0 = {k: v.copy() for k, v in mem.items()}
    out = pyref_format.format_to_optimize(path, cmds, speed, mem, 0.7, 0.4, dt, nb, n_poses=n, max_poses=max_poses, T=T)
    checked = 0
    for s in range(B):
        if n[s] < 1:
            assert out["T_scene"][s] == 0 and not out["robot_status"][s].any()
            continue
        have = mem0["valid"][s] != 0
        L = int(mem0["length"][s, 0])
        kept_s = max_poses - 1 if n[s] > max_poses else int(n[s])
        if have and kept_s - 1 > L:
            continue  # the reference indexes previous_cmds[i - 1] beyond its size for this combination (undefined behaviour)
        # (the adapter takes n command slots; the reference reads cmds[i - 1] for i < kept only, :537-545)
        ref = host_format(hostlib, path[s, :n[s]], cmds[s, :n[s]], mem0["prev_path"][s, :L] if have else None,
                          mem0["prev_cmds"][s, :L] if have else None, speed[s], 0.7, 0.4, max_time, dt)
        kept = out["T_scene"][s] + 1 if n[s] >= 2 else n[s]
        assert ref.shape[0] == kept, (s, n[s], ref.shape)
        assert np.max(np.abs(ref - out["robot_status"][s, :kept])) <= 1e-15, s
        assert not out["robot_status"][s, kept:].any()
        checked += 1
    assert checked >= 40
    assert out["T_scene"][:7].tolist() == [0, 0, 1, max_poses - 2, max_poses - 1, max_poses - 2, max_poses - 2]


@pytest.mark.gpu
def test_gpu_format_with_path_lengths_matches_pyref():
    from nav2_social_mpc_controller_amd.params import OptimizerParams
    from nav2_social_mpc_controller_amd.solver import BatchSolver
    from oracle import pyref_format
    prm = OptimizerParams.readme()
    s = BatchSolver(prm)
    max_poses = int(np.round(np.float32(prm.max_time) / np.float32(prm.time_step)))
    T, rows, B = max_poses - 1, max_poses + 1, 400
    nb = prm.dims(T, True)[2]
    path, cmds, speed, n = ragged_inputs(41, B, rows, max_poses)
    prev_path, prev_cmds, _ = make_inputs(42, B, T)
    rng = np.random.default_rng(43)
    mem = s.new_memory(B, T, lengths=True)
    plen = rng.integers(1, T + 2, size=B)
    for b in range(B):
        if b % 3:
            mem["prev_path"][b, :plen[b]], mem["prev_cmds"][b, :plen[b]] = prev_path[b, :plen[b]], prev_cmds[b, :plen[b]]
            mem["valid"][b], mem["length"][b] = 1, (plen[b], plen[b])
    mem_ref = {k: v.copy() for k, v in mem.items()}
    got = s.format_to_optimize(path, cmds, speed, mem, 0.7, 0.4, n_poses=n, max_poses=max_poses, T=T)
    exp = pyref_format.format_to_optimize(path, cmds, speed, mem_ref, 0.7, 0.4, prm.time_step, nb, n_poses=n,
                                          max_poses=max_poses, T=T)
    assert np.array_equal(got["T_scene"], exp["T_scene"])
    for k in ("robot_status", "pose0", "init_params", "path_pts", "goal_yaw"):
        err = np.abs(got[k] - exp[k])
        if k in ("robot_status", "pose0", "goal_yaw"):
            err = np.minimum(err, np.abs(err - 2 * np.pi))
        assert np.max(err) <= 1e-13, k
    for k in mem:
        assert np.array_equal(mem[k], mem_ref[k]), k
    # the store of a solve with those horizons, then a second format against records of mixed sizes
    status = rng.integers(0, 3, size=B).astype(np.int32)
    res_path, res_cmds, _ = make_inputs(44, B, T)
    s.memory_store(status, res_path, res_cmds, mem, T_scene=np.maximum(got["T_scene"], 1))
    pyref_format.memory_store(status, res_path, res_cmds, mem_ref, T_scene=np.maximum(exp["T_scene"], 1))
    for k in mem:
        assert np.array_equal(mem[k], mem_ref[k]), k
    path2, cmds2, speed2, n2 = ragged_inputs(45, B, rows, max_poses)
    got2 = s.format_to_optimize(path2, cmds2, speed2, mem, 0.7, 0.4, n_poses=n2, max_poses=max_poses, T=T)
    exp2 = pyref_format.format_to_optimize(path2, cmds2, speed2, mem_ref, 0.7, 0.4, prm.time_step, nb, n_poses=n2,
                                           max_poses=max_poses, T=T)
    assert np.array_equal(got2["T_scene"], exp2["T_scene"])
    err = np.abs(got2["robot_status"] - exp2["robot_status"])
    assert np.max(np.minimum(err, np.abs(err - 2 * np.pi))) <= 1e-13
