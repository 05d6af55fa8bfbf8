"""TEST INFRASTRUCTURE ONLY: plain-Python restatement of the geometric core of mpc::PathHandler::transformGlobalPlan,
SURVEY §8 row f4 (reference src/path_handler.cpp:39-108): the window of the global plan handed to the trajectorizer
and the pruning of what the robot has passed. PARITY UNPINNED (the reference holds no fixtures for it).
Third-party arithmetic restated from its published form (nav2_util/geometry_utils.hpp, ROS 2 Humble; the reference's
package.xml pins no version): euclidean_distance(pose, pose) = std::hypot(dx, dy) in 2-D,
first_after_integrated_distance (running sum of segment lengths, first element after the sum exceeds the bound),
min_by (first minimum, strict <). The tf2 transforms of the reference (:50-54, :74-88) are a rigid 2-D transform here
(plan frame -> costmap frame); ROS message plumbing is not restated."""
import math

import numpy as np

EMPTY_PLAN = 1        # "Received plan with zero length" (:44-47)
EMPTY_WINDOW = 2      # "Resulting plan has 0 poses in it." (:100-103)


def first_after_integrated_distance(pts, begin, end, bound):
    if begin == end:
        return end
    dist = 0.0
    for i in range(begin, end - 1):
        dist += math.hypot(pts[i + 1][0] - pts[i][0], pts[i + 1][1] - pts[i][1])
        if dist > bound:
            return i + 1
    return end


def min_by(values, begin, end):
    if begin == end:
        return end
    lowest, lowest_i = values(begin), begin
    for i in range(begin + 1, end):
        v = values(i)
        if v < lowest:
            lowest, lowest_i = v, i
    return lowest_i


def transform_global_plan(plan, start, robot_pose, max_robot_pose_search_dist, dist_threshold, to_local=None):
    """plan [n][2] (what setPlan stored), start = poses already erased by earlier calls, robot_pose (x, y, yaw) in the
    plan frame. Returns (window [m][2] in the costmap frame, new start, error)."""
    plan = np.asarray(plan, np.float64)
    n = plan.shape[0]
    if n - start <= 0:
        return np.zeros((0, 2)), start, EMPTY_PLAN
    rx, ry = float(robot_pose[0]), float(robot_pose[1])
    upper = first_after_integrated_distance(plan, start, n, max_robot_pose_search_dist)           # :56-59
    dist = lambda i: math.hypot(rx - plan[i][0], ry - plan[i][1])                                 # noqa: E731
    tb = min_by(dist, start, upper)                                                               # :61-66
    te = n
    for i in range(tb, n):                                                                        # :68-75
        if dist(i) > dist_threshold:
            te = i
            break
    win = plan[tb:te].copy()
    if to_local is not None:                                                                      # :77-96
        tx, ty, yaw = (float(v) for v in to_local)
        c, s = math.cos(yaw), math.sin(yaw)
        win = np.stack([tx + c * win[:, 0] - s * win[:, 1], ty + s * win[:, 0] + c * win[:, 1]], axis=1) if len(win) else win
    new_start = tb                                                                                # :98 (erase [begin, tb))
    if te - tb <= 0:
        return np.zeros((0, 2)), new_start, EMPTY_WINDOW
    return win, new_start, 0
