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


def test_host_fov_filter_matches_numpy_restatement(built):
    """The field-of-view filter of computeVelocityCommands (src/social_mpc_controller.cpp:196-214): C++ host mirror vs
    oracle/pyref_format.fov_filter."""
    import ctypes as C
    from oracle import pyref_format
    lib = C.CDLL(os.path.join(built, "libsmpc_host.so"))
    lib.smpc_host_fov_filter.restype = C.c_int
    lib.smpc_host_fov_filter.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_double, C.c_double, C.c_double, C.c_int, C.c_int,
                                         C.c_double, C.c_void_p]
    rng = np.random.default_rng(5)
    seen_in, seen_out = 0, 0
    for _ in range(40):
        pose = np.array([rng.uniform(3, 7), rng.uniform(3, 7), rng.uniform(-np.pi, np.pi)])
        n = 12
        r, phi = rng.uniform(0.5, 6.0, n), rng.uniform(-np.pi, np.pi, n)
        xy = np.ascontiguousarray(np.stack([pose[0] + r * np.cos(phi), pose[1] + r * np.sin(phi)], 1))
        keep = np.zeros(n, np.int32)
        k = lib.smpc_host_fov_filter(xy.ctypes.data, n, pose.ctypes.data, np.pi / 4, 0.0, 0.0, 200, 200, 0.05, keep.ctypes.data)
        people = np.zeros((n, 5)); people[:, :2] = xy
        want = pyref_format.fov_filter(people, n, pose, np.pi / 4, (0.0, 0.0), 200, 200, 0.05)
        assert k == len(want) and np.flatnonzero(keep).tolist() == want
        seen_in += k; seen_out += n - k
    assert seen_in > 20 and seen_out > 200


@pytest.mark.gpu
def test_controller_ticks_match_oracle(built, oracle, tmp_path):
    """Three control ticks through the host mirror of SocialMPCController::computeVelocityCommands (trajectorize ->
    field-of-view filter -> optimize -> first command); every tick's solve is replayed through the CPU oracle."""
    from nav2_social_mpc_controller_amd.params import OptimizerParams
    from nav2_social_mpc_controller_amd.scenes import SceneBatch
    prefix = str(tmp_path / "ctl")
    out = subprocess.check_output([os.path.join(built, "controller_demo"), "3", prefix], text=True)
    lines = [l for l in out.splitlines() if l.startswith("tick")]
    assert len(lines) == 3 and all("optimized=1" in l and "people_in_fov=1" in l for l in lines)   # 1 of 3 persons is seen
    prm = OptimizerParams.readme()
    for tick in range(3):
        d = _read_dump(f"{prefix}_{tick}.bin")
        assert (d["T"], d["N"], d["P"], d["hp"]) == (28, 3, 6, 1)      # 31 trajectorized poses cut to 29
        sc = SceneBatch(d["T"], d["N"], float(d["dt"]), d["pose0"][None], d["init"][None], d["pts"].reshape(1, -1, 2),
                        np.array([d["goal"]]), d["ppl"].reshape(1, d["T"] + 1, 6, d["N"]), np.array([d["hp"]], np.uint8),
                        d["cm"].reshape(1, d["sy"], d["sx"]), d["origin"][None], float(d["res"]), True)
        ref = oracle.solve(prm, sc, theta_zero_convention=True)
        assert ref["status"][0] == d["status"] and ref["iterations"][0] == d["iters"]
        assert np.max(np.abs(ref["cmds"][0].ravel() - d["cmds"])) <= 1e-5
        assert np.all(sc.people[0, :, 3, 1:] == -1.0)                   # two phantoms: only the person ahead passed the filter
        cmd = [float(v) for v in lines[tick].split("cmd=(")[1].split(")")[0].split(",")]
        assert abs(cmd[0] - d["cmds"][0]) <= 1e-9 and abs(cmd[1] - d["cmds"][1]) <= 1e-9   # the command returned = cmds[0]


def test_ros_plugin_shell_sources_keep_the_reference_interface():
    """host/ros/: the nav2_core::Controller plugin over the solver. ROS 2 / Nav2 are absent from this
    image, so the sources cannot be compiled here; what can be checked is that they declare the reference's interface
    (include/nav2_social_mpc_controller/social_mpc_controller.hpp:70-112) and export the reference's type string
    (src/social_mpc_controller.cpp:325, nav2_social_mpc_controller.xml), and that nothing in the default build needs them."""
    import re
    ros = os.path.join(HOST, "ros")
    hpp = open(os.path.join(ros, "social_mpc_controller_plugin.hpp")).read()
    cpp = open(os.path.join(ros, "social_mpc_controller_plugin.cpp")).read()
    xml = open(os.path.join(ros, "nav2_social_mpc_controller.xml")).read()
    cmake = open(os.path.join(ros, "CMakeLists.txt")).read()
    assert re.search(r"class SocialMPCController\s*:\s*public nav2_core::Controller", hpp)
    flat = re.sub(r"\s+", " ", hpp)
    for sig in ("void configure( const rclcpp_lifecycle::LifecycleNode::WeakPtr & parent, std::string name, "
                "std::shared_ptr<tf2_ros::Buffer> tf, std::shared_ptr<nav2_costmap_2d::Costmap2DROS> costmap_ros) override;",
                "void cleanup() override;", "void activate() override;", "void deactivate() override;",
                "geometry_msgs::msg::TwistStamped computeVelocityCommands( const geometry_msgs::msg::PoseStamped & pose, "
                "const geometry_msgs::msg::Twist & velocity, nav2_core::GoalChecker * goal_checker) override;",
                "void setPlan(const nav_msgs::msg::Path & path) override;",
                "void setSpeedLimit(const double & speed_limit, const bool & percentage) override;"):
        assert sig in flat, sig
    assert "PLUGINLIB_EXPORT_CLASS(nav2_social_mpc_controller::SocialMPCController, nav2_core::Controller)" in cpp
    assert 'type="nav2_social_mpc_controller::SocialMPCController"' in xml and 'base_class_type="nav2_core::Controller"' in xml
    assert 'library path="nav2_social_mpc_controller"' in xml
    assert "find_package(nav2_core QUIET)" in cmake and "SMPC_HOST_WITH_ROS" in cmake
    assert "#error" in hpp and "SMPC_HOST_WITH_ROS" in hpp          # refuses to build without ROS instead of faking it
    # the reference's parameter names (src/optimizer.cpp:26-84, src/path_trajectorizer.cpp:52-59, src/social_mpc_controller.cpp:58-60)
    for key in ("fov_angle", "transform_tolerance", "omnidirectional", "lookahead_dist", "max_angular_vel", "time_step", "max_time",
                "linear_solver_type", "param_tol", "fn_tol", "gradient_tol", "max_iterations", "debug_optimizer", "control_horizon",
                "parameter_block_length", "current_path_weight", "current_cmds_weight", "weights.distance_weight",
                "weights.social_weight", "weights.velocity_weight", "weights.angle_weight", "weights.agent_angle_weight",
                "weights.proxemics_weight", "weights.velocity_feasibility_weight", "weights.obstacle_weight",
                "weights.goal_align_weight"):
        assert ('"' + key + '"' in cpp) or ('.' + key + '"' in cpp), key
    assert not os.path.exists(os.path.join(ros, "ros_stubs")) and "host/ros" not in open(os.path.join(HOST, "Makefile")).read()
