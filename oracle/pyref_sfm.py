"""Independent numpy restatement of the people projection that precedes the hot path:
Optimizer::project_people + computeObstacle (reference src/optimizer.cpp:554-728) with the Social Force Model of
include/nav2_social_mpc_controller/sfm.hpp (computeForces :462-485, updatePosition :525-551; group forces are zero
because project_people never sets a groupId). TEST INFRASTRUCTURE ONLY (checker of the HIP kernel and of the C++ host
adapter). PARITY UNPINNED like the rest of the oracle (no reference fixtures, reference not buildable here)."""
import math

import numpy as np

F_DESIRED, F_OBSTACLE, SIGMA_OBSTACLE, F_SOCIAL, LAMBDA, GAMMA, N_, NPRIME, RELAX = 2.0, 20.0, 0.2, 2.1, 2.0, 0.35, 2.0, 3.0, 0.5


def _wrap(a):
    while a <= -math.pi:
        a += 2 * math.pi
    while a > math.pi:
        a -= 2 * math.pi
    return a


def _normalized(v):
    z = v[0] * v[0] + v[1] * v[1]
    return v / math.sqrt(z) if z > 0 else v


class GridError(RuntimeError):
    pass


def compute_obstacle(pos, od):
    """src/optimizer.cpp:673-728 — returns agent - obstacle (a DIFFERENCE), float arithmetic as in the reference."""
    w, h, res = od["width"], od["height"], np.float32(od["resolution"])
    if od["indexes"].size == 0 or w <= 0 or h <= 0 or res <= 0:
        raise GridError("invalid grid")
    xc = math.floor((pos[0] - od["origin_x"]) / float(res))
    yc = math.floor((pos[1] - od["origin_y"]) / float(res))
    xc = int(xc) & 0xFFFFFFFF if xc >= 0 else (int(xc) + (1 << 32))   # (unsigned int) conversion
    yc = int(yc) & 0xFFFFFFFF if yc >= 0 else (int(yc) + (1 << 32))
    if xc >= w or yc >= h:
        raise GridError("cell out of bounds")
    ob = int(od["indexes"].reshape(-1)[xc + yc * w])
    if ob >= w * h:
        raise GridError("index out of bounds")
    oy, ox = ob // w, ob % w
    x = np.float32(float(np.float32(ox) * res) + od["origin_x"])
    y = np.float32(float(np.float32(oy) * res) + od["origin_y"])
    return np.array([pos[0] - float(x), pos[1] - float(y)])


def project_people(init_people, robot_path, od, maxtime, timestep, theta_zero_convention=False):
    """init_people [N][6], robot_path [T+1][6] -> people_proj [T+1][N][6] (fields x,y,yaw,t,lv,av).
    theta_zero_convention: take theta == 0 exactly when both velocities are exactly equal (two standing people), where
    the reference's value is libm last-bit noise — the convention of the HIP kernel."""
    maxtime, timestep = np.float32(maxtime), np.float32(timestep)
    N = init_people.shape[0]
    traj = [init_people.copy()]
    agents = []
    for i in range(N):
        if init_people[i, 3] == -1:
            continue
        yaw, lv = init_people[i, 2], init_people[i, 4]
        a = dict(pos=init_people[i, :2].copy(), yaw=yaw, lv=lv, av=init_people[i, 5],
                 vel=np.array([lv * math.cos(yaw), lv * math.sin(yaw)]), des=0.5, radius=0.5)
        a["goal"] = (a["pos"] + float(maxtime) * a["vel"], 0.25)
        if od["width"] == 100 and od["height"] == 100:      # "NOT valid" grid: the person is dropped (:598-603)
            continue
        a["obst"] = [compute_obstacle(a["pos"], od)]
        agents.append(a)
    dt = float(timestep)
    for i in range(robot_path.shape[0] - 1):
        r = robot_path[i]
        robot = dict(pos=r[:2].copy(), yaw=r[2], lv=r[4], av=r[5], vel=np.array([r[4] * math.cos(r[2]), r[4] * math.sin(r[2])]),
                     des=0.6, radius=0.5, goal=(robot_path[-1, :2].copy(), 0.25), obst=[])
        agents.append(robot)
        forces = []
        for idx, me in enumerate(agents):
            if me["goal"] is not None and np.linalg.norm(me["goal"][0] - me["pos"]) > me["goal"][1]:
                d = _normalized(me["goal"][0] - me["pos"])
                desired = F_DESIRED * (d * me["des"] - me["vel"]) / RELAX
            else:
                desired = -me["vel"] / RELAX
            obstacle = np.zeros(2)
            if me["obst"]:
                for o in me["obst"]:
                    md = me["pos"] - o            # the stored difference is used as a position (reference quirk)
                    dist = np.linalg.norm(md) - me["radius"]
                    obstacle = obstacle + F_OBSTACLE * math.exp(-dist / SIGMA_OBSTACLE) * _normalized(md)
                obstacle = obstacle / len(me["obst"])
            social = np.zeros(2)
            for j, other in enumerate(agents):
                if j == idx:
                    continue
                diff = other["pos"] - me["pos"]
                dd = _normalized(diff)
                iv = LAMBDA * (me["vel"] - other["vel"]) + dd
                il = math.sqrt(iv[0] ** 2 + iv[1] ** 2)
                idir = iv / il
                theta = _wrap(_wrap(math.atan2(dd[1], dd[0])) - _wrap(math.atan2(idir[1], idir[0])))
                if theta_zero_convention and LAMBDA * (me["vel"][0] - other["vel"][0]) == 0.0 and LAMBDA * (me["vel"][1] - other["vel"][1]) == 0.0:
                    theta = 0.0
                B = GAMMA * il
                nd = math.sqrt(diff[0] ** 2 + diff[1] ** 2)
                fv = -math.exp(-nd / B - (NPRIME * B * theta) ** 2)
                sign = 0.0 if theta == 0 else (1.0 if theta > 0 else -1.0)      # sfm.hpp:265-270
                fa = -sign * math.exp(-nd / B - (N_ * B * theta) ** 2)
                social = social + F_SOCIAL * (fv * idir + fa * np.array([-idir[1], idir[0]]))
            forces.append(desired + social + obstacle)
        for me, f in zip(agents, forces):
            me["vel"] = me["vel"] + f * dt
            sp = math.sqrt(me["vel"][0] ** 2 + me["vel"][1] ** 2)
            if sp > me["des"]:
                me["vel"] = _normalized(me["vel"]) * me["des"]
            init_yaw = me["yaw"]
            me["yaw"] = _wrap(math.atan2(me["vel"][1], me["vel"][0]))
            me["av"] = _wrap(me["yaw"] - init_yaw) / dt
            me["pos"] = me["pos"] + me["vel"] * dt
            me["lv"] = math.sqrt(me["vel"][0] ** 2 + me["vel"][1] ** 2)
            if me["goal"] is not None and np.linalg.norm(me["goal"][0] - me["pos"]) <= me["goal"][1]:
                me["goal"] = None
        agents.pop()
        for a in agents:
            a["obst"] = [compute_obstacle(a["pos"], od)]
        step = np.zeros((N, 6))
        step[:, 3] = -1.0
        for k, a in enumerate(agents):
            step[k] = [a["pos"][0], a["pos"][1], a["yaw"], float(np.float32(i + 1) * timestep), a["lv"], a["av"]]
        traj.append(step)
    return np.stack(traj)
