"""TEST INFRASTRUCTURE ONLY: numpy / plain-Python restatement of PathTrajectorizer::trajectorize, SURVEY §8 row f3
(reference src/path_trajectorizer.cpp:120-288, motion model path_trajectorizer.hpp:106-135). PARITY UNPINNED (the
reference holds no fixtures for it; cross-checked against the independent C++ restatement
nav2_social_mpc_controller_amd/host/path_trajectorizer.cpp by tests/test_trajectorize.py).
Third-party arithmetic: angles::normalize_angle (ros/angles, unpinned) in its ROS 2 form."""
import math

import numpy as np


def normalize_angle(a):
    r = math.fmod(a + math.pi, 2.0 * math.pi)
    return r + math.pi if r <= 0.0 else r - math.pi


def yaw_roundtrip(yaw):
    sz, cz = math.sin(yaw * 0.5), math.cos(yaw * 0.5)
    return math.atan2(2.0 * (cz * sz), cz * cz - sz * sz)


def trajectorize(plan, robot_pose, omnidirectional=False, desired_linear_vel=0.4, lookahead_dist=0.4,
                 max_angular_vel=1.0, time_step=0.05, max_time=3.0):
    """plan [n][2], robot_pose (x, y, yaw). Returns (path [m][3], cmds [m-1][3] (vx, vy, wz), error) or (None, None, 1)
    when the reference returns false. error 2: no way-point candidate (the reference reads poses[-1])."""
    plan = np.asarray(plan, np.float64)
    if plan.shape[0] < 2:
        return None, None, 1
    max_steps = int(np.round(max_time / time_step))
    rx, ry, rth = float(robot_pose[0]), float(robot_pose[1]), float(robot_pose[2])
    path = [(rx, ry, rth)]
    cmds = []
    goal_dist, steps, err = 1000.0, 0, 0
    while goal_dist > 0.2 and steps < max_steps:
        wp_index, min_dist = -1, 100.0
        for i in range(plan.shape[0] - 1, -1, -1):
            wx, wy = plan[i]
            d = math.sqrt((rx - wx) * (rx - wx) + (ry - wy) * (ry - wy))
            if d <= lookahead_dist:
                wp_index = i
                break
            if d < min_dist:
                min_dist, wp_index = d, i
        if wp_index < 0:
            err = 2
            break
        wpx, wpy = plan[wp_index]
        dx = (wpx - rx) * math.cos(rth) + (wpy - ry) * math.sin(rth)
        dy = -(wpx - rx) * math.sin(rth) + (wpy - ry) * math.cos(rth)
        dtheta = normalize_angle(math.atan2(dy, dx))
        vx = vy = wz = 0.0
        if omnidirectional:
            vx = desired_linear_vel * math.cos(dtheta)
            vy = desired_linear_vel * math.sin(dtheta)
        else:
            pd2 = dx * dx + dy * dy
            curvature = 2.0 * dy / pd2 if pd2 > 0.001 else 0.0
            vx = desired_linear_vel
            if abs(dtheta) > math.pi / 2.0:
                vx = 0.0
                wz = max_angular_vel * (1.0 if dtheta > 0 else -1.0)
            else:
                wz = vx * curvature
        rx, ry, rth = (rx + (vx * math.cos(rth) + vy * math.cos(math.pi / 2 + rth)) * time_step,
                       ry + (vx * math.sin(rth) + vy * math.sin(math.pi / 2 + rth)) * time_step,
                       rth + wz * time_step)
        path.append((rx, ry, yaw_roundtrip(rth)))
        cmds.append((vx, vy, wz))
        gx, gy = plan[-1]
        goal_dist = math.sqrt((rx - gx) * (rx - gx) + (ry - gy) * (ry - gy))
        steps += 1
    return np.array(path), np.array(cmds).reshape(-1, 3), err
