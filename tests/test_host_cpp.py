"""The C++ host adapter (nav2_social_mpc_controller_amd/host): the reference's Optimizer interface over the C ABI."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "nav2_social_mpc_controller_amd", "host")


@pytest.fixture(scope="module")
def built():
    assert os.path.exists(os.path.join(ROOT, "nav2_social_mpc_controller_amd", "csrc", "libsmpc_hip.so")), "run __graft_entry__.build()"
    subprocess.check_call(["make", "-C", HOST, "-s"])
    return HOST


def test_host_rows_restated_from_the_reference(built):
    """people_to_status / format_to_optimize / project_people / computeObstacle quirks (CPU only)."""
    out = subprocess.check_output([os.path.join(built, "host_cpu_tests")], text=True)
    assert "all checks passed" in out


@pytest.mark.skipif(os.path.exists("/dev/kfd"), reason="checks the no-GPU failure mode")
def test_optimizer_initialize_fails_loudly_without_gpu(built):
    r = subprocess.run([os.path.join(built, "host_demo"), "1"], capture_output=True, text=True)
    assert r.returncode == 2 and "no HIP device" in r.stderr


def _read_dump(fn):
    raw = open(fn, "rb").read()
    hdr = np.frombuffer(raw, np.int32, 8)
    T, N, P, sx, sy, hp, status, iters = (int(v) for v in hdr)
    off = 32
    def take(n):
        nonlocal off
        a = np.frombuffer(raw, np.float64, n, off).copy()
        off += 8 * n
        return a
    dt, res, goal = take(3)
    pose0, origin = take(3), take(2)
    init, pts, ppl = take(P), take(2 * (T + 1)), take((T + 1) * 6 * N)
    cmds, path = take(2 * (T + 1)), take(3 * (T + 1))
    cm = np.frombuffer(raw, np.uint8, sx * sy, off).copy()
    return dict(T=T, N=N, P=P, sx=sx, sy=sy, hp=hp, status=status, iters=iters, dt=dt, res=res, goal=goal, pose0=pose0,
                origin=origin, init=init, pts=pts, ppl=ppl, cmds=cmds, path=path, cm=cm)


@pytest.mark.gpu
def test_optimize_closed_loop_matches_oracle(built, oracle, tmp_path):
    """Three closed-loop control ticks through Optimizer::optimize (warm-start memory, SFM people projection on the
    host, the solve on the GPU); every tick's C-ABI inputs are replayed through the CPU oracle."""
    from nav2_social_mpc_controller_amd.params import OptimizerParams
    from nav2_social_mpc_controller_amd.scenes import SceneBatch
    prefix = str(tmp_path / "tick")
    out = subprocess.check_output([os.path.join(built, "host_demo"), "3", prefix], text=True)
    assert out.count("ok=1") == 3
    prm = OptimizerParams.readme()
    for tick in range(3):
        d = _read_dump(f"{prefix}_{tick}.bin")
        assert (d["T"], d["N"], d["P"]) == (28, 3, 6)          # 40 poses cut to 29 (src/optimizer.cpp:492-497)
        sc = SceneBatch(d["T"], d["N"], float(d["dt"]), d["pose0"][None], d["init"][None], d["pts"].reshape(1, -1, 2),
                        np.array([d["goal"]]), d["ppl"].reshape(1, d["T"] + 1, 6, d["N"]), np.array([d["hp"]], np.uint8),
                        d["cm"].reshape(1, d["sy"], d["sx"]), d["origin"][None], float(d["res"]), True)
        ref = oracle.solve(prm, sc, theta_zero_convention=True)
        assert ref["status"][0] == d["status"] and ref["iterations"][0] == d["iters"]
        assert np.max(np.abs(ref["cmds"][0].ravel() - d["cmds"])) <= 1e-5
        assert np.max(np.abs(ref["path"][0][:, :2].ravel() - d["path"].reshape(-1, 3)[:, :2].ravel())) <= 1e-5
        # third agent is a phantom (2 people in the demo), padded exactly like people_to_status does
        assert np.all(sc.people[0, :, 3, 2] == -1.0)
