"""Closed-loop batch episodes (nav2_social_mpc_controller_amd/episode.py): every tick chains the device versions of
format_to_optimize (f2), project_people (f1) and the solve (a1-a12), then stores the TrajectoryMemory. Each stage of each
tick is replayed on the CPU from the inputs the device saw (teacher forcing, so errors do not compound over ticks)."""
import numpy as np
import pytest

CMD_TOL = 1e-5  # north_star tolerance on the optimised command sequence


@pytest.mark.gpu
def test_episode_ticks_match_cpu_chain(oracle):
    from nav2_social_mpc_controller_amd.episode import BatchEpisode
    from nav2_social_mpc_controller_amd.params import OptimizerParams
    from nav2_social_mpc_controller_amd.scenes import SceneBatch, make_scenes, uniform
    from oracle import pyref_format, pyref_sfm

    prm = OptimizerParams.readme()
    B, N = 48, 3
    sc = make_scenes(prm, B, N, n_valid=2)
    T = sc.T
    CH, bl, nb, P, M, _ = prm.dims(T, True)
    w_ref = (uniform(0x5EED0001, np.arange(B), 6)[:, 0] * 2.0 - 1.0) * 0.6
    cells, res, origin = 480, float(np.float32(0.1)), np.array([-16.0, -16.0])
    od_idx = np.zeros((cells, cells), np.uint32)
    ep = BatchEpisode(prm, sc, w_ref, od_idx, origin, res)
    od = dict(width=cells, height=cells, resolution=res, origin_x=origin[0], origin_y=origin[1], indexes=od_idx)
    n_checked = 0
    for tick in range(3):
        r = ep.tick(record=True)
        # the command handed to the robot: cmds[0] of the solve (no fallback is active in these scenes)
        assert (ep.cmd_source.cpu().numpy() == 0).all() and np.array_equal(ep.cmd_vel.cpu().numpy(), r.result["cmds"][:, 0])
        # --- f2: people_to_status (no field-of-view filter in this episode)
        st, has = pyref_format.people_to_status(r.persons, r.person_count, N)
        assert np.max(np.abs(r.init_people - st)) <= 1e-14 and np.array_equal(r.has_people, has)
        # --- f2: format_to_optimize + memory initialisation
        mem = {k: v.copy() for k, v in r.memory_before.items()}
        exp = pyref_format.format_to_optimize(r.plan_path, r.plan_cmds, r.speed, mem, prm.current_path_weight,
                                              prm.current_cmds_weight, prm.time_step, nb)
        for k in ("robot_status", "pose0", "init_params", "path_pts", "goal_yaw"):
            err = np.abs(getattr(r, k) - exp[k])
            err = np.minimum(err, np.abs(err - 2 * np.pi))
            assert np.max(err) <= 1e-13, (tick, k)
        assert (r.memory_before["valid"] == (0 if tick == 0 else 1)).all()
        assert (r.T_scene == T).all()   # no global plans: every robot has the batch's horizon
        # --- f1: project_people from the formatted robot status (a few scenes: the numpy restatement is slow)
        assert (r.proj_error == 0).all()
        for s in range(0, B, 12):
            pp = pyref_sfm.project_people(r.init_people[s], r.robot_status[s], od, prm.max_time, prm.time_step,
                                          theta_zero_convention=True)          # [T+1][N][6]
            got = r.people_proj[s].transpose(0, 2, 1)                           # [T+1][6][N] -> [T+1][N][6]
            assert np.max(np.abs(got - pp)) <= 1e-9, (tick, s)
        # --- a1-a12: the solve on exactly the inputs the device assembled
        scene = SceneBatch(T, N, prm.dt, r.pose0, r.init_params, r.path_pts, r.goal_yaw,
                           np.ascontiguousarray(r.people_proj), r.has_people, sc.costmap, sc.costmap_origin,
                           sc.resolution, False)
        rz = oracle.solve(prm, scene, nthreads=8, theta_zero_convention=True)
        firm = rz["marginal_decisions"] == 0
        assert firm.mean() >= 0.8
        err = np.max(np.abs(r.result["cmds"] - rz["cmds"]).reshape(B, -1), axis=1)
        assert np.max(err[firm]) <= CMD_TOL, (tick, float(np.max(err[firm])))
        assert np.array_equal(r.result["status"][firm], rz["status"][firm])
        n_checked += int(firm.sum())
        # --- memory store: usable solves overwrite their record with the optimised path / cmds
        ok = r.result["status"] != 2
        assert np.array_equal(r.memory_after["prev_path"][ok], r.result["path"][ok])
        assert np.array_equal(r.memory_after["prev_cmds"][ok], r.result["cmds"][ok])
        assert (r.memory_after["valid"] == 1).all()
    assert n_checked >= 100
    # the robots actually moved: closed loop, not three solves of the same scene
    assert np.max(np.abs(ep.pose.cpu().numpy()[:, :2] - sc.pose0[:, :2])) > 0.02


@pytest.mark.gpu
def test_episode_with_trajectorizer_matches_cpu_chain(oracle):
    """Plan mode: trajectorize (f3) -> format (f2, cut to T + 1 of the max_steps + 1 poses) -> project (f1) -> solve."""
    from nav2_social_mpc_controller_amd.episode import BatchEpisode, arc_plans
    from nav2_social_mpc_controller_amd.params import OptimizerParams, TrajectorizerParams
    from nav2_social_mpc_controller_amd.scenes import SceneBatch, make_scenes, uniform
    from oracle import pyref_format, pyref_trajectorize

    prm = OptimizerParams.readme()
    tp = TrajectorizerParams(desired_linear_vel=0.6, lookahead_dist=0.4, max_angular_vel=1.0, time_step=0.05, max_time=1.5)
    B, N = 32, 3
    sc = make_scenes(prm, B, N, n_valid=2)
    T = sc.T + 1                                                   # plan mode: sized for a path of exactly max_poses poses
    max_poses = tp.max_steps
    assert T == max_poses - 1 and prm.rollout_steps == max_poses - 2
    CH, bl, nb, P, M, _ = prm.dims(T, True)
    w_ref = (uniform(0x5EED0001, np.arange(B), 6)[:, 0] * 2.0 - 1.0) * 0.6
    plan, plan_len = arc_plans(sc.pose0, 0.4 * w_ref)  # radius >= 4.2 m: the 20 m arc never closes on itself
    fov = 1.2  # wider than the reference default (pi/4) so that a good share of the scenes keeps somebody in view
    ep = BatchEpisode(prm, sc, w_ref, np.zeros((480, 480), np.uint32), np.array([-16.0, -16.0]), float(np.float32(0.1)),
                      plan=plan, plan_len=plan_len, traj_params=tp, fov_angle=fov)
    seen = 0
    for tick in range(3):
        r = ep.tick(record=True)
        # the command handed to the robot: cmds[0] of the solve (no fallback is active in these scenes)
        assert (ep.cmd_source.cpu().numpy() == 0).all() and np.array_equal(ep.cmd_vel.cpu().numpy(), r.result["cmds"][:, 0])
        assert (r.traj_n_poses == tp.max_steps + 1).all()
        # f4 / f2: field-of-view filter + people_to_status from the world people and the pose the tick started from
        for s in range(B):
            keep = pyref_format.fov_filter(r.persons[s], r.person_count[s], r.robot_pose[s], fov, sc.costmap_origin[s],
                                           sc.size_x, sc.size_y, sc.resolution)
            sel = r.persons[s][keep][None] if keep else np.zeros((1, 1, 5))
            st, has = pyref_format.people_to_status(sel, np.array([len(keep)]), N)
            assert np.max(np.abs(r.init_people[s] - st[0])) <= 1e-14 and r.has_people[s] == has[0], (tick, s)
            seen += len(keep)
        for s in range(0, B, 5):  # f3 on the pose the tick started from
            p, c, err = pyref_trajectorize.trajectorize(plan[s], r.robot_pose[s], tp.omnidirectional, tp.desired_linear_vel,
                                                        tp.lookahead_dist, tp.max_angular_vel, tp.time_step, tp.max_time)
            assert err == 0
            dyaw = np.abs(r.plan_path[s, :, 2] - p[:, 2])
            assert np.max(np.abs(r.plan_path[s, :, :2] - p[:, :2])) <= 1e-11 and np.max(np.minimum(dyaw, np.abs(dyaw - 2 * np.pi))) <= 1e-11
            assert np.max(np.abs(r.plan_cmds[s, :-1] - c[:, [0, 2]])) <= 1e-11
        mem = {k: v.copy() for k, v in r.memory_before.items()}
        exp = pyref_format.format_to_optimize(r.plan_path, r.plan_cmds, r.speed, mem, prm.current_path_weight,
                                              prm.current_cmds_weight, prm.time_step, nb, n_poses=r.traj_n_poses,
                                              max_poses=max_poses, T=T)
        for k in ("robot_status", "pose0", "init_params", "path_pts", "goal_yaw"):
            err = np.abs(getattr(r, k) - exp[k])
            assert np.max(np.minimum(err, np.abs(err - 2 * np.pi))) <= 1e-13, (tick, k)
        # full-length paths (max_steps + 1 poses) are cut to max_poses - 1 poses: the horizon of README's T = 28
        assert np.array_equal(r.T_scene, exp["T_scene"]) and (r.T_scene == prm.rollout_steps).all()
        scene = SceneBatch(T, N, prm.dt, r.pose0, r.init_params, r.path_pts, r.goal_yaw,
                           np.ascontiguousarray(r.people_proj), r.has_people, sc.costmap, sc.costmap_origin,
                           sc.resolution, False, r.T_scene)
        rz = oracle.solve(prm, scene, nthreads=8, theta_zero_convention=True)
        firm = rz["marginal_decisions"] == 0
        err = np.max(np.abs(r.result["cmds"] - rz["cmds"]).reshape(B, -1), axis=1)
        assert firm.mean() >= 0.8 and np.max(err[firm]) <= CMD_TOL, (tick, float(np.max(err[firm])))
    assert np.max(np.abs(ep.pose.cpu().numpy()[:, :2] - sc.pose0[:, :2])) > 0.02
    assert 0 < seen < 3 * B * 2   # the filter kept some persons and dropped some


@pytest.mark.gpu
def test_short_plans_are_solved_on_their_own_horizon_and_reach_the_goal(oracle):
    """The reference solves whatever horizon the tick produces (T = optim_velocities.size(), src/optimizer.cpp:237,
    248-249) and falls back only when the solve is unusable (src/social_mpc_controller.cpp:241-245): a robot whose
    trajectorized path ends inside the batch's horizon is solved with its own T_b, and keeps driving on MPC commands
    until the trajectorizer stops producing steps at the goal."""
    from nav2_social_mpc_controller_amd.episode import BatchEpisode, arc_plans
    from nav2_social_mpc_controller_amd.params import OptimizerParams, TrajectorizerParams
    from nav2_social_mpc_controller_amd.scenes import SceneBatch, make_scenes, uniform
    from oracle import pyref_format

    prm = OptimizerParams.readme()
    tp = TrajectorizerParams(desired_linear_vel=0.6, lookahead_dist=0.4, max_angular_vel=1.0, time_step=0.05, max_time=1.5)
    B, N = 32, 3
    sc = make_scenes(prm, B, N, n_valid=2)
    w_ref = (uniform(0x5EED0001, np.arange(B), 6)[:, 0] * 2.0 - 1.0) * 0.6
    long_plan, plan_len = arc_plans(sc.pose0, 0.4 * w_ref)
    short = np.arange(B) % 2 == 1
    plan_len = np.where(short, 12, plan_len).astype(np.int32)     # 12 poses x 0.05 m = 0.55 m: reached in < T steps
    ep = BatchEpisode(prm, sc, w_ref, np.zeros((480, 480), np.uint32), np.array([-16.0, -16.0]), float(np.float32(0.1)),
                      plan=long_plan, plan_len=plan_len, traj_params=tp, fov_angle=1.2)
    T, max_poses = ep.T, tp.max_steps
    nb = prm.dims(T, True)[2]
    goal = long_plan[np.arange(B), plan_len - 1]
    d0 = np.hypot(*(sc.pose0[:, :2] - goal).T)
    dmin = d0.copy()
    horizons = []
    for tick in range(30):
        r = ep.tick(record=True)
        src = ep.cmd_source.cpu().numpy()
        assert (src == 0).all(), (tick, src)                       # MPC commands all the way, for short paths too
        assert (r.result["status"] != 2).all()
        assert np.array_equal(ep.cmd_vel.cpu().numpy(), r.result["cmds"][:, 0])
        mem = {k: v.copy() for k, v in r.memory_before.items()}
        exp = pyref_format.format_to_optimize(r.plan_path, r.plan_cmds, r.speed, mem, prm.current_path_weight,
                                              prm.current_cmds_weight, prm.time_step, nb, n_poses=r.traj_n_poses,
                                              max_poses=max_poses, T=T)
        assert np.array_equal(r.T_scene, exp["T_scene"])
        assert (r.T_scene[~short] == prm.rollout_steps).all() and r.T_scene.min() >= 1
        # on its way to the goal a short-plan robot has a short horizon (once it has arrived the goal checker of the
        # controller server would stop calling the plugin; here the loop just keeps ticking)
        d = np.hypot(*(r.robot_pose[:, :2] - goal).T)
        on_the_way = short & (d > 0.3) & (dmin > 0.3)
        assert (r.T_scene[on_the_way] < prm.rollout_steps).all(), (tick, r.T_scene[on_the_way])
        dmin = np.minimum(dmin, d)
        horizons.append(np.where(on_the_way, r.T_scene, -1))
        if tick in (0, 7, 15, 29):   # the solve of every robot against the oracle run with the robot's own horizon
            scene = SceneBatch(T, N, prm.dt, r.pose0, r.init_params, r.path_pts, r.goal_yaw,
                               np.ascontiguousarray(r.people_proj), r.has_people, sc.costmap, sc.costmap_origin,
                               sc.resolution, False, r.T_scene)
            rz = oracle.solve(prm, scene, nthreads=8, theta_zero_convention=True)
            firm = rz["marginal_decisions"] == 0
            err = np.max(np.abs(r.result["cmds"] - rz["cmds"]).reshape(B, -1), axis=1)
            # (short horizons end on tiny costs, where many decisions sit inside the rounding noise the oracle flags)
            assert firm.mean() >= 0.5 and np.max(err[firm]) <= CMD_TOL, (tick, float(firm.mean()), float(np.max(err[firm])))
            assert np.array_equal(r.result["iterations"][firm], rz["iterations"][firm])
            assert (r.result["status"][~firm] != 2).all()
        # the record of a usable solve holds T_b + 1 poses and commands
        assert np.array_equal(r.memory_after["length"], np.stack([r.T_scene + 1, r.T_scene + 1], axis=1))
    horizons = np.array(horizons)
    assert (horizons[0][short] > 1).all()
    # the trajectorizer stops within 0.2 m of the plan's end (src/path_trajectorizer.cpp:150-152): the robots that the MPC
    # lets drive (one that stands beside a person may wait there for the whole test) arrive on optimised commands, and
    # their horizon shrinks on the way
    arrived = short & (dmin < 0.3)
    assert arrived.sum() >= 0.6 * short.sum(), (int(arrived.sum()), dmin[short])
    for b in np.where(arrived)[0]:
        h = horizons[:, b][horizons[:, b] >= 0]
        assert len(h) >= 3 and h[-1] < h[0], (b, h)


@pytest.mark.gpu
def test_sharded_episode_equals_the_single_chain():
    """The robots of a batch do not interact: four shards on four HIP streams must return, robot by robot, the very
    numbers of the one-stream episode (three ticks, people + global plans + field-of-view filter)."""
    from nav2_social_mpc_controller_amd.episode import BatchEpisode, ShardedEpisode, arc_plans
    from nav2_social_mpc_controller_amd.params import OptimizerParams, TrajectorizerParams
    from nav2_social_mpc_controller_amd.scenes import make_scenes, uniform

    prm = OptimizerParams.readme()
    tp = TrajectorizerParams(desired_linear_vel=0.6, lookahead_dist=0.4, max_angular_vel=1.0, time_step=0.05, max_time=1.5)
    B, N = 203, 8   # not a multiple of the shard count
    sc = make_scenes(prm, B, N)
    w_ref = (uniform(0x5EED0001, np.arange(B), 6)[:, 0] * 2.0 - 1.0) * 0.6
    plan, plan_len = arc_plans(sc.pose0, 0.4 * w_ref)
    od = (np.zeros((480, 480), np.uint32), np.array([-16.0, -16.0]), float(np.float32(0.1)))
    kw = dict(plan=plan, plan_len=plan_len, traj_params=tp, fov_angle=2.0, plan_window=(4.0, 10.0))  # pruning state included
    one = BatchEpisode(prm, sc, w_ref, *od, **kw)
    four = ShardedEpisode(prm, sc, w_ref, *od, shards=4, graphs=True, **kw)  # every shard's tick replayed from a HIP graph
    assert [p.B for p in four.parts] == [50, 51, 51, 51]
    for _ in range(3):
        one.tick()
        four.tick()
    one.synchronize()
    for name in ("status", "iterations", "cmds", "path", "final_cost"):
        assert np.array_equal(one.res[name].cpu().numpy(), four.gather(name).cpu().numpy()), name
    for name in ("pose", "cmd_vel", "cmd_source", "proj_error", "plan_start", "window_len"):
        assert np.array_equal(getattr(one, name).cpu().numpy(), four.gather(name).cpu().numpy()), name
    # a shard can still be ticked outside its graph (recording / per-stage timing): same numbers again
    one.tick()
    r = four.parts[1].tick(record=True)
    one.synchronize()
    sl = four.slices[1]
    assert np.array_equal(r.result["cmds"], one.res["cmds"].cpu().numpy()[sl])


@pytest.mark.gpu
def test_episode_with_the_plan_window_of_the_path_handler():
    """computeVelocityCommands' order (src/social_mpc_controller.cpp:171-180): transformGlobalPlan, then trajectorize on
    the window. Replayed on the CPU from what the device saw: window and pruning against pyref_path_handler, the
    trajectorized path against pyref_trajectorize on that window."""
    from nav2_social_mpc_controller_amd.episode import BatchEpisode, arc_plans
    from nav2_social_mpc_controller_amd.params import OptimizerParams, TrajectorizerParams
    from nav2_social_mpc_controller_amd.scenes import make_scenes, uniform
    from oracle import pyref_path_handler, pyref_trajectorize

    prm = OptimizerParams.readme()
    tp = TrajectorizerParams(desired_linear_vel=0.6, lookahead_dist=0.4, max_angular_vel=1.0, time_step=0.05, max_time=1.5)
    B, N = 24, 3
    sc = make_scenes(prm, B, N, n_valid=2)
    w_ref = (uniform(0x5EED0001, np.arange(B), 6)[:, 0] * 2.0 - 1.0) * 0.6
    plan, plan_len = arc_plans(sc.pose0, 0.4 * w_ref)
    plan_len = plan_len.copy()
    lost = np.zeros(B, bool)
    lost[[5, 17]] = True
    plan_len[5] = 0                  # "Received plan with zero length" (src/path_handler.cpp:44-47)
    plan[17] += 40.0                 # a plan far from the robot: "Resulting plan has 0 poses in it." (:100-103)
    search, thr = 2.0, 3.0
    ep = BatchEpisode(prm, sc, w_ref, np.zeros((480, 480), np.uint32), np.array([-16.0, -16.0]), float(np.float32(0.1)),
                      plan=plan, plan_len=plan_len, traj_params=tp, fov_angle=1.2, plan_window=(search, thr))
    start = np.zeros(B, np.int32)
    pose_lost = sc.pose0[lost].copy()
    for tick in range(8):
        r = ep.tick(record=True)
        # a robot whose transformGlobalPlan throws gets NO command in that cycle (the exception leaves
        # computeVelocityCommands before the trajectorizer and its 0.1 m/s fallback): source 3, it does not move
        src = ep.cmd_source.cpu().numpy()
        assert (r.window_err[lost] != 0).all() and (r.window_err[~lost] == 0).all()
        assert (src[lost] == 3).all() and not ep.cmd_vel.cpu().numpy()[lost].any()
        assert np.array_equal(ep.pose.cpu().numpy()[lost], pose_lost)
        for s in np.where(~lost)[0]:
            win, ns, err = pyref_path_handler.transform_global_plan(plan[s, :plan_len[s]], int(start[s]), r.robot_pose[s], search, thr)
            assert err == 0 and r.plan_start[s] == ns and r.window_len[s] == len(win), (tick, s)
            assert np.array_equal(r.window[s, :len(win)], win)
            if s % 4 == 0:
                p, c, terr = pyref_trajectorize.trajectorize(win, r.robot_pose[s], tp.omnidirectional, tp.desired_linear_vel,
                                                             tp.lookahead_dist, tp.max_angular_vel, tp.time_step, tp.max_time)
                assert terr == 0 and r.traj_n_poses[s] == p.shape[0]
                assert np.max(np.abs(r.plan_path[s, :p.shape[0], :2] - p[:, :2])) <= 1e-11
        start = r.plan_start.copy()
        assert (src[~lost] == 0).all()
    assert start.max() >= 1 and (r.window_len[~lost] < plan_len[~lost]).all()   # the plans were pruned and clipped


@pytest.mark.gpu
def test_concurrent_streams_returns_distinct_streams():
    import torch

    from nav2_social_mpc_controller_amd.episode import concurrent_streams
    for n in (1, 3, 4):
        st = concurrent_streams(n, "cuda:0")
        assert len(st) == n and len({s.cuda_stream for s in st}) == n
        assert all(s.cuda_stream != torch.cuda.default_stream("cuda:0").cuda_stream for s in st)
