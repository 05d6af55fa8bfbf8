"""TEST INFRASTRUCTURE ONLY (see oracle/README or DESIGN.md §oracle): numpy restatement of the reference's warm-start
formatting, SURVEY §8 row f2. PARITY UNPINNED (the reference holds no fixtures for it; cross-checked against the
independent C++ restatement in nav2_social_mpc_controller_amd/host/optimizer.cpp by tests/test_format.py).

Follows, per scene:
  Optimizer::optimize, memory initialisation      reference src/optimizer.cpp:172-186
  Optimizer::format_to_optimize                   reference src/optimizer.cpp:484-551
  what optimize() derives from optim_status       reference src/optimizer.cpp:197-261, 298
  TrajectoryMemory store                          reference src/optimizer.cpp:448-449 (skipped on an unusable solve, :384-388)
"""
import math

import numpy as np


def yaw_roundtrip(yaw: float) -> float:
    """tf2::Quaternion::setRPY(0, 0, yaw) -> toMsg -> tf2::getYaw (x = y = 0)."""
    sz, cz = math.sin(yaw * 0.5), math.cos(yaw * 0.5)
    return math.atan2(2.0 * (cz * sz), cz * cz - sz * sz)


def new_memory(B, T, lengths=False):
    m = {"prev_path": np.zeros((B, T + 1, 3)), "prev_cmds": np.zeros((B, T + 1, 2)), "valid": np.zeros(B, np.int32)}
    if lengths:  # previous_path.poses.size(), previous_cmds.size() per record (scenes with horizons of their own)
        m["length"] = np.zeros((B, 2), np.int32)
    return m


def format_to_optimize(path, cmds, speed, memory, current_path_w, current_cmds_w, time_step, nb, n_poses=None,
                       max_poses=0, T=None):
    """path [B,rows,3], cmds [B,rows,2], speed [B,2]; memory updated in place like the singleton. Plain loops.
    n_poses [B]: poses of each incoming path (its commands: one fewer, one per trajectorizer step); max_poses =
    round(max_time / time_step): a longer path is cut to max_poses - 1 poses (:491-497). T: stride of the outputs
    (default rows - 1); rows a scene does not have stay zero. Without n_poses every path has T + 1 poses."""
    path = np.asarray(path, np.float64)
    cmds = np.asarray(cmds, np.float64)
    B, rows, _ = path.shape
    T = rows - 1 if T is None else int(T)
    Tp = T + 1
    wp = float(np.float32(current_path_w))
    wc = float(np.float32(current_cmds_w))
    ts = np.float32(time_step)
    out = {"robot_status": np.zeros((B, Tp, 6)), "pose0": np.zeros((B, 3)), "init_params": np.zeros((B, 2 * nb)),
           "path_pts": np.zeros((B, Tp, 2)), "goal_yaw": np.zeros(B), "T_scene": np.zeros(B, np.int32)}
    for s in range(B):
        n = Tp if n_poses is None else max(int(n_poses[s]), 0)
        ncmd = Tp if n_poses is None else max(n - 1, 0)
        if not memory["valid"][s]:  # :177-183: previous := the whole incoming path / commands (what the record can hold)
            keep_p, keep_c = min(n, Tp), min(ncmd, Tp)
            memory["prev_path"][s, :keep_p] = path[s, :keep_p]
            memory["prev_cmds"][s, :keep_c] = cmds[s, :keep_c]
            memory["valid"][s] = 1
            if "length" in memory:
                memory["length"][s] = (keep_p, keep_c)
        plen, clen = (memory["length"][s] if "length" in memory else (Tp, Tp))
        pp, pc = memory["prev_path"][s], memory["prev_cmds"][s]
        kept = n
        if max_poses > 0 and n > max_poses:  # :491-497
            kept = max_poses - 1
        kept = min(kept, Tp)
        out["T_scene"][s] = max(kept - 1, 0)
        for i in range(kept):
            x, y, yaw = path[s, i]
            if i < plen:  # "!previous_path.poses.empty() && i < previous_path.poses.size()" (:504)
                x = wp * path[s, i, 0] + (1.0 - wp) * pp[i, 0]
                y = wp * path[s, i, 1] + (1.0 - wp) * pp[i, 1]
                yaw = yaw_roundtrip(wp * path[s, i, 2] + (1.0 - wp) * pp[i, 2])
            if i == 0:
                lv, av = speed[s, 0], speed[s, 1]
            elif i - 1 < clen:
                lv = wc * cmds[s, i - 1, 0] + (1.0 - wc) * pc[i - 1, 0]
                av = wc * cmds[s, i - 1, 1] + (1.0 - wc) * pc[i - 1, 1]
            else:  # the reference reads previous_cmds out of bounds here (undefined): the current command alone
                lv = wc * cmds[s, i - 1, 0] + (1.0 - wc) * cmds[s, i - 1, 0]
                av = wc * cmds[s, i - 1, 1] + (1.0 - wc) * cmds[s, i - 1, 1]
            out["robot_status"][s, i] = (x, y, yaw, float(np.float32(i) * ts), lv, av)
        st = out["robot_status"][s]
        if kept > 0:
            out["pose0"][s] = (st[0, 0], st[0, 1], yaw_roundtrip(st[0, 2]))
            out["goal_yaw"][s] = st[kept - 1, 2]
        for b in range(min(nb, kept)):  # parameter blocks alias optim_velocities[0..nb-1] (:254-261)
            out["init_params"][s, 2 * b:2 * b + 2] = st[b, 4:6]
        out["path_pts"][s, :kept] = st[:kept, 0:2]
    return out


def memory_store(status, path, cmds, memory, T_scene=None):
    for s in range(len(status)):
        if status[s] != 2:
            n = path.shape[1] if T_scene is None else int(T_scene[s]) + 1
            memory["prev_path"][s, :n] = path[s, :n]
            memory["prev_cmds"][s, :n] = cmds[s, :n]
            memory["valid"][s] = 1
            if "length" in memory:
                memory["length"][s] = (n, n)


def people_to_status(people, count, N=3):
    """Optimizer::people_to_status (reference src/optimizer.cpp:454-482): people [B][Np][5] (px, py, vx, vy, vz)."""
    B = people.shape[0]
    out = np.zeros((B, N, 6))
    out[:, :, 3] = -1.0
    for s in range(B):
        for a in range(min(int(count[s]), N, people.shape[1])):
            px, py, vx, vy, vz = people[s, a]
            out[s, a] = (px, py, math.atan2(vy, vx), 0.0, math.sqrt(vx * vx + vy * vy), vz)
    return out, (np.asarray(count) != 0).astype(np.uint8)


def fov_filter(people, count, robot_pose, fov_angle, origin, size_x, size_y, resolution):
    """The field-of-view filter of SocialMPCController::computeVelocityCommands (reference
    src/social_mpc_controller.cpp:196-214) for one scene: returns the indices of the persons that are kept.
    Third-party arithmetic restated: nav2_costmap_2d Costmap2D::worldToMap, angles::shortest_angular_distance (ROS 2)."""
    keep = []
    robot_yaw = np.float32(robot_pose[2])
    for a in range(int(count)):
        px, py = people[a, 0], people[a, 1]
        if px < origin[0] or py < origin[1]:
            continue
        mx, my = int((px - origin[0]) / resolution), int((py - origin[1]) / resolution)
        if not (mx < size_x and my < size_y):
            continue
        angle = np.float32(math.atan2(py - robot_pose[1], px - robot_pose[0]))
        r = math.fmod((float(angle) - float(robot_yaw)) + math.pi, 2.0 * math.pi)
        rel = np.float32(r + math.pi if r <= 0.0 else r - math.pi)
        if float(abs(rel)) < fov_angle:
            keep.append(a)
    return keep
