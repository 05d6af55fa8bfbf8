"""Closed-loop (receding-horizon) batch episodes: B independent robots, every tick runs the whole
`Optimizer::optimize` chain of the reference on the device,

    field-of-view filter + people_to_status (src/social_mpc_controller.cpp:196-214, src/optimizer.cpp:454-482)
                                                                                   -> smpc_people_to_status_batch
    format_to_optimize + TrajectoryMemory   (src/optimizer.cpp:172-190, 484-551)  -> smpc_format_to_optimize_batch
    project_people                           (src/optimizer.cpp:554-728)           -> smpc_project_people_batch
    problem assembly + ceres::Solve + unpack (src/optimizer.cpp:197-446)           -> smpc_solve_batch
    memory store                             (src/optimizer.cpp:448-449)           -> smpc_memory_store_batch

preceded, when global plans are given, by PathTrajectorizer::trajectorize (src/path_trajectorizer.cpp:120-288 ->
smpc_trajectorize_path_batch) as in SocialMPCController::computeVelocityCommands (src/social_mpc_controller.cpp:176-189),
with all state resident in HBM (torch tensors are only the allocator here). What the reference gets from outside is
stood in for by the simplest thing that has the same shape:
  * without global plans: a constant-curvature arc from the current pose in place of the trajectorizer output;
  * the world: the robot executes the command computeVelocityCommands returns for one period — the first optimised
    command (it lands on the first pose of the optimised path), the trajectorizer's first command when the solve was
    not usable (src/social_mpc_controller.cpp:241-245), 0.1 m/s straight ahead when there is no trajectory (:180-189);
    people move with constant velocity.
"""
import ctypes as C
from dataclasses import dataclass

import numpy as np

from ._abi import (SmpcFormatBatch, SmpcFormatOut, SmpcMemoryBatch, SmpcPeopleBatch, SmpcProjectionBatch, SmpcSceneBatch,
                   SmpcTrajectorizeOut, SmpcPlanWindowBatch)
from .params import OptimizerParams, TrajectorizerParams
from .scenes import SceneBatch
from .solver import BatchSolver


@dataclass
class TickRecord:
    """Host copies of one tick's intermediate results (only filled when `record=True`; tests replay them on the CPU)."""
    plan_path: np.ndarray
    plan_cmds: np.ndarray
    speed: np.ndarray
    init_people: np.ndarray
    memory_before: dict
    robot_status: np.ndarray
    pose0: np.ndarray
    init_params: np.ndarray
    path_pts: np.ndarray
    goal_yaw: np.ndarray
    people_proj: np.ndarray
    proj_error: np.ndarray
    result: dict
    memory_after: dict
    robot_pose: np.ndarray = None   # [B,3] pose the tick started from
    traj_n_poses: np.ndarray = None
    window: np.ndarray = None       # [B,L,2] plan window handed to the trajectorizer (plan_window episodes)
    window_len: np.ndarray = None
    plan_start: np.ndarray = None   # [B] pruned start of every plan after this tick's window
    persons: np.ndarray = None      # [B,Np,5] world people (px, py, vx, vy, vz) the tick started from
    person_count: np.ndarray = None
    has_people: np.ndarray = None
    T_scene: np.ndarray = None      # [B] horizon every robot was solved with (plan episodes: from its own path length)
    window_err: np.ndarray = None   # [B] smpc_window_error of the tick's transformGlobalPlan (plan_window episodes)


class BatchEpisode:
    def __init__(self, params: OptimizerParams, scenes: SceneBatch, w_ref: np.ndarray, od_indexes: np.ndarray,
                 od_origin: np.ndarray, od_resolution: float, device: int = 0, plan: np.ndarray = None,
                 plan_len: np.ndarray = None, traj_params: TrajectorizerParams = None, fov_angle: float = None,
                 order_hint: bool = False, plan_window: tuple = None):
        """scenes: the start state (pose0, people at step 0, costmaps); w_ref [B]: curvature of the arc stand-in;
        od_*: one ObstacleDistance grid shared by all scenes. plan [B,L,2] + plan_len [B] + traj_params: global plans,
        trajectorized on the device every tick (the plan must stay longer than the horizon for the whole episode).
        fov_angle: field-of-view half angle of the people filter (reference default pi/4); None = no filter.
        plan_window: (max_robot_pose_search_dist, dist_threshold): every tick starts with PathHandler::transformGlobalPlan
        (smpc_transform_global_plan_batch; plan frame = costmap frame) and trajectorizes the window instead of the whole
        plan, as computeVelocityCommands does (src/social_mpc_controller.cpp:171-180; the reference passes 4.0 m and half
        the costmap's larger side); the plans are pruned as the robots
        advance. None: the global plans go to the trajectorizer as they are.
        order_hint: hand the solve kernel's queue the scenes sorted by the previous tick's sweep counts, longest first
        (smpc_scene_batch.order; the results are the same, the lone launch is shorter)."""
        import torch

        self.torch = torch
        self.params = params
        self.solver = BatchSolver(params, device)
        self.dev = f"cuda:{device}"
        # library kernels and the few torch ops of the world model share one stream, so they are ordered
        self.solver.set_stream(torch.cuda.current_stream(self.dev).cuda_stream)
        B, T, N = scenes.B, scenes.T, scenes.N
        if plan is not None:
            # Plan mode: every robot is solved with the horizon its own trajectorized path gives it (smpc_format_batch.n_poses
            # -> T_scene). The arrays are sized for the longest one: a path of exactly max_poses = round(max_time / dt)
            # poses is not cut by format_to_optimize (src/optimizer.cpp:491-497) and so has one step more than the longer,
            # cut, paths: T = max_poses - 1 = rollout_steps + 1.
            assert T == params.rollout_steps
            T = T + 1
        self.B, self.T, self.N = B, T, N
        CH, bl, nb, P, M, _ = params.dims(T, True)
        self.P = P
        f64 = dict(dtype=torch.float64, device=self.dev)
        self.pose = torch.from_numpy(scenes.pose0.copy()).to(self.dev)                 # [B,3] current robot pose
        self.speed = torch.from_numpy(scenes.init_params[:, 0:2].copy()).to(self.dev)  # [B,2] current twist
        self.w_ref = torch.from_numpy(np.ascontiguousarray(w_ref, np.float64)).to(self.dev)
        # world people as people_msgs::Person rows [B,Np,5]: position x, y, velocity x, y, z (scene people at step 0;
        # make_scenes keeps the invalid ones at the end, so the first `count` rows are the persons)
        st0 = scenes.people[:, 0].transpose(0, 2, 1)                                    # [B,N,6] x, y, yaw, t, lv, av
        persons = np.stack([st0[:, :, 0], st0[:, :, 1], st0[:, :, 4] * np.cos(st0[:, :, 2]), st0[:, :, 4] * np.sin(st0[:, :, 2]),
                            st0[:, :, 5]], axis=-1)
        count = np.where(scenes.has_people != 0, (st0[:, :, 3] != -1.0).sum(axis=1), 0).astype(np.int32)
        self.persons = torch.from_numpy(np.ascontiguousarray(persons)).to(self.dev)
        self.person_count = torch.from_numpy(count).to(self.dev)
        self.fov_angle = fov_angle
        self.people = torch.zeros((B, N, 6), **f64)                                     # people_to_status output
        self.has_people = torch.zeros(B, dtype=torch.uint8, device=self.dev)
        self.costmap = torch.from_numpy(scenes.costmap).to(self.dev)
        self.costmap_origin = torch.from_numpy(scenes.costmap_origin).to(self.dev)
        self.costmap_shared = scenes.costmap_shared
        self.size_x, self.size_y, self.resolution = scenes.size_x, scenes.size_y, scenes.resolution
        self.od_indexes = torch.from_numpy(np.ascontiguousarray(od_indexes, np.uint32).view(np.int32)).to(self.dev)
        self.od_origin = torch.from_numpy(np.ascontiguousarray(od_origin, np.float64).reshape(1, 2)).to(self.dev)
        self.od_h, self.od_w = int(od_indexes.shape[-2]), int(od_indexes.shape[-1])
        self.od_resolution = float(od_resolution)
        # TrajectoryMemory, one record per scene
        self.mem_path = torch.zeros((B, T + 1, 3), **f64)
        self.mem_cmds = torch.zeros((B, T + 1, 2), **f64)
        self.mem_valid = torch.zeros(B, dtype=torch.int32, device=self.dev)
        self.mem_length = torch.zeros((B, 2), dtype=torch.int32, device=self.dev)   # poses / commands of every record
        self.T_scene = torch.full((B,), T, dtype=torch.int32, device=self.dev)      # horizon of every robot this tick
        self.max_poses = int(np.round(np.float32(params.max_time) / np.float32(params.time_step)))
        # per-tick buffers
        self.traj = traj_params
        if plan is not None:
            assert traj_params is not None and traj_params.max_steps + 1 >= T + 1
            self.plan = torch.from_numpy(np.ascontiguousarray(plan, np.float64)).to(self.dev)
            self.plan_len = torch.from_numpy(np.ascontiguousarray(plan_len, np.int32)).to(self.dev)
            self.rows = traj_params.max_steps + 1
            self.traj_n = torch.zeros(B, dtype=torch.int32, device=self.dev)
            self.traj_err = torch.zeros(B, dtype=torch.int32, device=self.dev)
            self.traj_vy = torch.zeros((B, self.rows), **f64)
            self.plan_window = plan_window
            if plan_window is not None:
                self.plan_start = torch.zeros(B, dtype=torch.int32, device=self.dev)
                self.window = torch.zeros_like(self.plan)
                self.window_len = torch.zeros(B, dtype=torch.int32, device=self.dev)
                self.window_err = torch.zeros(B, dtype=torch.int32, device=self.dev)
        else:
            self.plan = None
            self.plan_window = None
            self.rows = T + 1
        self.plan_path = torch.zeros((B, self.rows, 3), **f64)
        self.plan_cmds = torch.zeros((B, self.rows, 2), **f64)
        self.robot_status = torch.zeros((B, T + 1, 6), **f64)
        self.pose0 = torch.zeros((B, 3), **f64)
        self.init_params = torch.zeros((B, P), **f64)
        self.path_pts = torch.zeros((B, T + 1, 2), **f64)
        self.goal_yaw = torch.zeros(B, **f64)
        self.people_proj = torch.zeros((B, T + 1, 6, N), **f64)
        self.proj_error = torch.zeros(B, dtype=torch.int32, device=self.dev)
        self.rb, self.res = self.solver.alloc_results(B, T, self.dev)
        self.cmd_vel = torch.zeros((B, 2), **f64)                      # the command returned to the robot this tick
        self.cmd_source = torch.zeros(B, dtype=torch.int32, device=self.dev)  # 0 optimised, 1 trajectorizer, 2 creep, 3 none
        self.order_hint = order_hint
        self.graph = None
        self.gstream = None
        # queue order for the next solve (from the last solve's sweep counts; index order before the first one)
        self.order = torch.arange(B, dtype=torch.int32, device=self.dev)
        self.ticks = 0

    # -- trajectorizer (row f3) on the global plans, or the arc stand-in: v = 0.6, w = w_ref from the current pose --
    def _plan(self, timing: dict = None):
        torch = self.torch
        if self.plan is not None:
            tb = self.solver.trajectorize_c(self.traj, self.B, int(self.plan.shape[1]), 1)
            tb.plan, tb.plan_len, tb.robot_pose = self.plan.data_ptr(), self.plan_len.data_ptr(), self.pose.data_ptr()
            if self.plan_window is not None:  # the trajectorizer sees the window of the plan around the robot
                wb = SmpcPlanWindowBatch()
                wb.B, wb.L, wb.on_device = self.B, int(self.plan.shape[1]), 1
                wb.max_robot_pose_search_dist, wb.dist_threshold = float(self.plan_window[0]), float(self.plan_window[1])
                wb.plan, wb.plan_len, wb.plan_start = self.plan.data_ptr(), self.plan_len.data_ptr(), self.plan_start.data_ptr()
                wb.robot_pose = self.pose.data_ptr()
                self.solver.transform_global_plan_device(wb, self.window.data_ptr(), self.window_len.data_ptr(), self.window_err.data_ptr())
                if timing is not None:
                    timing["window_ms"] = self.solver.last_kernel_ms()
                tb.plan, tb.plan_len = self.window.data_ptr(), self.window_len.data_ptr()
            to = SmpcTrajectorizeOut()
            to.path, to.cmds, to.cmds_vy = self.plan_path.data_ptr(), self.plan_cmds.data_ptr(), self.traj_vy.data_ptr()
            to.n_poses, to.error = self.traj_n.data_ptr(), self.traj_err.data_ptr()
            self.solver.trajectorize_device(tb, to)
            return
        dt = self.params.dt
        k = torch.arange(self.T + 1, dtype=torch.float64, device=self.dev)[None, :]
        th = self.pose[:, 2:3] + self.w_ref[:, None] * dt * k
        inc = 0.6 * dt * torch.stack([torch.cos(th), torch.sin(th)], dim=-1)           # step k -> k+1
        xy = self.pose[:, None, 0:2] + torch.cumsum(inc, dim=1) - inc                  # exclusive prefix sum
        self.plan_path[:, :, 0:2] = xy
        self.plan_path[:, :, 2] = th
        self.plan_cmds[:, :, 0] = 0.6
        self.plan_cmds[:, :, 1] = self.w_ref[:, None]

    def _memory_c(self) -> SmpcMemoryBatch:
        mb = SmpcMemoryBatch()
        mb.prev_path, mb.prev_cmds, mb.valid = self.mem_path.data_ptr(), self.mem_cmds.data_ptr(), self.mem_valid.data_ptr()
        mb.length = self.mem_length.data_ptr()
        return mb

    def _memory_host(self) -> dict:
        return {"prev_path": self.mem_path.cpu().numpy().copy(), "prev_cmds": self.mem_cmds.cpu().numpy().copy(),
                "valid": self.mem_valid.cpu().numpy().copy(), "length": self.mem_length.cpu().numpy().copy()}

    def tick(self, record: bool = False, timing: dict = None):
        """One controller period for all B robots. Returns a TickRecord when `record`, else None. `timing`: a dict
        that receives the HIP-event duration (ms) of each stage's kernel (synchronises after every stage)."""
        torch = self.torch
        if self.gstream is not None and torch.cuda.current_stream(self.dev) != self.gstream:
            with torch.cuda.stream(self.gstream):  # after capture_graph the library's handle lives on the capture stream
                return self.tick(record, timing)
        s, prm, B, T, N = self.solver, self.params, self.B, self.T, self.N
        pose_before = self.pose.cpu().numpy().copy() if record else None
        self._plan(timing)
        if timing is not None and self.plan is not None:
            timing["trajectorize_ms"] = s.last_kernel_ms()
        rec = {}
        if record:
            if self.plan is not None:
                rec.update(traj_n_poses=self.traj_n.cpu().numpy().copy())
                if self.plan_window is not None:
                    rec.update(window=self.window.cpu().numpy().copy(), window_len=self.window_len.cpu().numpy().copy(),
                               plan_start=self.plan_start.cpu().numpy().copy())
            rec.update(robot_pose=pose_before, persons=self.persons.cpu().numpy().copy(),
                       person_count=self.person_count.cpu().numpy().copy())
        # 0. field-of-view filter + people_to_status
        qb = SmpcPeopleBatch()
        qb.B, qb.Np, qb.N, qb.on_device = B, int(self.persons.shape[1]), N, 1
        qb.people, qb.count = self.persons.data_ptr(), self.person_count.data_ptr()
        if self.fov_angle is not None:
            qb.robot_pose, qb.fov_angle = self.pose.data_ptr(), float(self.fov_angle)
            qb.costmap_origin, qb.costmap_shared = self.costmap_origin.data_ptr(), 1 if self.costmap_shared else 0
            qb.size_x, qb.size_y, qb.resolution = self.size_x, self.size_y, self.resolution
        s.people_to_status_device(qb, self.people.data_ptr(), self.has_people.data_ptr())
        if timing is not None:
            timing["people_ms"] = s.last_kernel_ms()
        if record:
            rec.update(plan_path=self.plan_path.cpu().numpy().copy(), plan_cmds=self.plan_cmds.cpu().numpy().copy(),
                       speed=self.speed.cpu().numpy().copy(), init_people=self.people.cpu().numpy().copy(),
                       has_people=self.has_people.cpu().numpy().copy(), memory_before=self._memory_host())
        # 1. format_to_optimize + memory
        fb = SmpcFormatBatch()
        fb.B, fb.T, fb.path_rows, fb.on_device = B, T, self.rows, 1
        fb.time_step = float(prm.dt)
        fb.current_path_w, fb.current_cmds_w = float(prm.current_path_weight), float(prm.current_cmds_weight)
        fb.path, fb.cmds, fb.speed = self.plan_path.data_ptr(), self.plan_cmds.data_ptr(), self.speed.data_ptr()
        fb.memory = self._memory_c()
        if self.plan is not None:  # every robot with the horizon of its own trajectorized path
            fb.n_poses, fb.max_poses = self.traj_n.data_ptr(), self.max_poses
        fo = SmpcFormatOut()
        fo.robot_status, fo.pose0, fo.init_params = self.robot_status.data_ptr(), self.pose0.data_ptr(), self.init_params.data_ptr()
        fo.path_pts, fo.goal_yaw = self.path_pts.data_ptr(), self.goal_yaw.data_ptr()
        fo.T_scene = self.T_scene.data_ptr()
        s.format_device(fb, fo)
        if timing is not None:
            timing["format_ms"] = s.last_kernel_ms()
        # 2. project_people
        pb = SmpcProjectionBatch()
        pb.B, pb.T, pb.N, pb.on_device = B, T, N, 1
        pb.max_time, pb.time_step = float(prm.max_time), float(prm.time_step)
        pb.init_people, pb.robot_path = self.people.data_ptr(), self.robot_status.data_ptr()
        pb.od_indexes, pb.od_shared = self.od_indexes.data_ptr(), 1
        pb.od_width, pb.od_height, pb.od_resolution = self.od_w, self.od_h, self.od_resolution
        pb.od_origin = self.od_origin.data_ptr()
        s.project_people_device(pb, self.people_proj.data_ptr(), self.proj_error.data_ptr())
        if timing is not None:
            timing["project_ms"] = s.last_kernel_ms()
        # 3. solve
        sb = SmpcSceneBatch()
        sb.B, sb.T, sb.N, sb.on_device = B, T, N, 1
        sb.dt = prm.dt
        sb.pose0, sb.init_params, sb.path_pts = self.pose0.data_ptr(), self.init_params.data_ptr(), self.path_pts.data_ptr()
        sb.goal_yaw, sb.people, sb.has_people = self.goal_yaw.data_ptr(), self.people_proj.data_ptr(), self.has_people.data_ptr()
        sb.costmap, sb.costmap_shared = self.costmap.data_ptr(), 1 if self.costmap_shared else 0
        sb.size_x, sb.size_y = self.size_x, self.size_y
        sb.costmap_origin, sb.resolution = self.costmap_origin.data_ptr(), self.resolution
        if self.order_hint:
            sb.order = self.order.data_ptr()  # longest scenes of the previous period first
        if self.plan is not None:
            sb.T_scene = self.T_scene.data_ptr()
        s.solve_device(sb, self.rb)
        if timing is not None:
            timing["solve_ms"] = s.last_kernel_ms()
        if self.order_hint:
            self.order.copy_(BatchSolver.longest_first(self.res["evaluations"]))
        # 4. memory store (usable solves only; T_scene + 1 poses and commands of each)
        mb = self._memory_c()
        s.memory_store_device(B, T, self.res["status"].data_ptr(), self.res["path"].data_ptr(), self.res["cmds"].data_ptr(), mb,
                              self.T_scene.data_ptr() if self.plan is not None else 0)
        if timing is not None:
            timing["store_ms"] = s.last_kernel_ms()
        if record:
            rec.update(T_scene=self.T_scene.cpu().numpy().copy())
            if self.plan_window is not None:
                rec.update(window_err=self.window_err.cpu().numpy().copy())
            rec.update(robot_status=self.robot_status.cpu().numpy().copy(), pose0=self.pose0.cpu().numpy().copy(),
                       init_params=self.init_params.cpu().numpy().copy(), path_pts=self.path_pts.cpu().numpy().copy(),
                       goal_yaw=self.goal_yaw.cpu().numpy().copy(), people_proj=self.people_proj.cpu().numpy().copy(),
                       proj_error=self.proj_error.cpu().numpy().copy(),
                       result={k: v.cpu().numpy().copy() for k, v in self.res.items()}, memory_after=self._memory_host())
        # 5. the command computeVelocityCommands returns (fallbacks included), then the world moves one period with it:
        #    a usable solve lands the robot on the first optimised pose, a fallback command is integrated with the
        #    trajectorizer's own motion model (x, y with the old heading, then the heading)
        s.select_command_device(B, T, self.rows, self.traj_n.data_ptr() if self.plan is not None else 0,
                                self.plan_cmds.data_ptr(), self.res["status"].data_ptr(), self.res["cmds"].data_ptr(),
                                self.cmd_vel.data_ptr(), self.cmd_source.data_ptr(),
                                self.window_err.data_ptr() if self.plan_window is not None else 0)
        dt = prm.dt
        v, w, th = self.cmd_vel[:, 0], self.cmd_vel[:, 1], self.pose[:, 2]
        moved = torch.stack([self.pose[:, 0] + v * torch.cos(th) * dt, self.pose[:, 1] + v * torch.sin(th) * dt, th + w * dt], dim=1)
        optimised = (self.cmd_source == 0)[:, None]
        # state is updated in place: every buffer the kernels read keeps its address from tick to tick (capture_graph)
        self.pose.copy_(torch.where(optimised, self.res["path"][:, 0, :], moved))
        self.speed.copy_(self.cmd_vel)
        self.persons[:, :, 0] += self.persons[:, :, 2] * dt
        self.persons[:, :, 1] += self.persons[:, :, 3] * dt
        self.ticks += 1
        return TickRecord(**rec) if record else None

    def capture_graph(self, stream=None):
        """Record one tick (every kernel of the chain and the torch ops of the world model) into a HIP graph on `stream`
        (a new side stream by default) and make tick() replay it: one graph launch per control period instead of ~25
        kernel launches from Python, which is what lets several independent shards keep the GPU busy (ShardedEpisode).
        One warm-up tick runs first (the library's buffers grow on first use); the episode's state is put back after it."""
        torch = self.torch
        self.gstream = stream if stream is not None else torch.cuda.Stream(device=self.dev)
        self.gstream.wait_stream(torch.cuda.current_stream(self.dev))
        self.solver.set_stream(self.gstream.cuda_stream)
        state = ("pose", "speed", "persons", "mem_path", "mem_cmds", "mem_valid", "mem_length", "order") + (("plan_start",) if self.plan_window is not None else ())
        with torch.cuda.stream(self.gstream):
            saved = {k: getattr(self, k).clone() for k in state}
            ticks = self.ticks
            self.tick()  # warm-up on the capture stream: the library's buffers grow, torch's allocator settles
        self.gstream.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph, stream=self.gstream):
            self.tick()
        with torch.cuda.stream(self.gstream):  # the world is where it was before the warm-up tick
            for k in state:
                getattr(self, k).copy_(saved[k])
        self.ticks = ticks
        return self.graph

    def replay(self):
        """One control period through the captured graph (on the capture stream)."""
        with self.torch.cuda.stream(self.gstream):
            self.graph.replay()
        self.ticks += 1

    def synchronize(self):
        self.torch.cuda.synchronize()


def concurrent_streams(n: int, device: str, candidates: int = 12):
    """n torch streams that really run side by side. Streams are mapped onto a few hardware queues by the runtime; two
    streams that share a queue execute one after the other, and which pool stream lands where is not visible from
    here (three shards were measured at 2.7 ms or 4.1 ms per tick depending on nothing but the streams the pool
    handed out). Candidates are probed with a pair of spin kernels: a pair that takes as long as the two spins in a
    row shares a queue. Falls back to the first n pool streams when the probe kernel is not available."""
    import time

    import torch

    pool = [torch.cuda.Stream(device=device) for _ in range(max(n, candidates))]
    spin = getattr(torch.cuda, "_sleep", None)
    if spin is None or n < 2:
        return pool[:n]
    cycles = 1_000_000  # a few hundred microseconds

    def pair_time(a, b):
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        for st in ((a,) if b is None else (a, b)):
            with torch.cuda.stream(st):
                spin(cycles)
        torch.cuda.synchronize(device)
        return time.perf_counter() - t0

    pair_time(pool[0], pool[1])  # warm-up
    alone = min(pair_time(pool[0], None) for _ in range(3))
    chosen = [pool[0]]
    for cand in pool[1:]:
        if len(chosen) == n:
            break
        if all(min(pair_time(c, cand) for _ in range(2)) < 1.5 * alone for c in chosen):
            chosen.append(cand)
    for cand in pool:  # not enough independent queues: fill up with whatever is left
        if len(chosen) == n:
            break
        if cand not in chosen:
            chosen.append(cand)
    return chosen


class ShardedEpisode:
    """The robots of a batch are independent of each other (every scene has its own plan, people, memory), so a tick of
    B robots can be issued as `shards` chains of B / shards robots on separate HIP streams: while one shard's solve
    launch drains its last long-running scenes, the other shards' kernels keep the CUs busy (the same overlap the
    serving bench gets from independent batches; a single chain pays the tail of every launch). Three things make it
    pay (measured at 8192 robots, 8 people, tools/gpu_episode.py / tools/shard_caps.sh): every shard's tick is one HIP
    graph launch (from Python the ~25 launches per shard-tick make the host the bottleneck); every solve launch sizes its
    persistent grid to 1 / shards of the resident wavefronts (smpc_set_solve_share: oversubscribed grids leave the other
    shards' small kernels waiting for wave slots); the small kernels run at the highest wave priority (next to another
    shard's solve they took 5-10x longer at equal priority). Three shards: 2.72 ms per tick against 3.24-3.32 ms for
    the single chain; two: 2.87; four and more fall behind again (more streams than hardware queues). Per-robot results
    are those of the one-stream episode bit for bit (kernels take every decision per scene / per lane)."""

    def __init__(self, params: OptimizerParams, scenes: SceneBatch, w_ref: np.ndarray, od_indexes: np.ndarray,
                 od_origin: np.ndarray, od_resolution: float, device: int = 0, plan: np.ndarray = None,
                 plan_len: np.ndarray = None, traj_params: TrajectorizerParams = None, fov_angle: float = None,
                 shards: int = 3, order_hint: bool = False, graphs: bool = True, solve_share: int = None,
                 plan_window: tuple = None):
        import torch

        self.torch = torch
        B = scenes.B
        shards = max(1, min(shards, B))
        edges = [B * k // shards for k in range(shards + 1)]
        self.slices = [slice(edges[k], edges[k + 1]) for k in range(shards) if edges[k + 1] > edges[k]]
        self.streams = concurrent_streams(len(self.slices), f"cuda:{device}")
        self.parts = []
        for sl, st in zip(self.slices, self.streams):
            idx = np.arange(sl.start, sl.stop)
            with torch.cuda.stream(st):  # the shard's solver handle binds to the stream current at construction
                self.parts.append(BatchEpisode(
                    params, scenes.select(idx), np.asarray(w_ref)[idx], od_indexes, od_origin, od_resolution, device=device,
                    plan=None if plan is None else plan[idx], plan_len=None if plan_len is None else plan_len[idx],
                    traj_params=traj_params, fov_angle=fov_angle, order_hint=order_hint, plan_window=plan_window))
        self.B = B
        self.graphs = graphs
        for part in self.parts:  # every shard's persistent solve grid takes its share of the resident wavefronts
            part.solver.set_solve_share(solve_share if solve_share else len(self.parts))
        if graphs:  # one HIP graph per shard: a tick is `shards` graph launches
            for part, st in zip(self.parts, self.streams):
                part.capture_graph(st)

    def tick(self):
        for part, st in zip(self.parts, self.streams):
            if self.graphs:
                part.replay()
            else:
                with self.torch.cuda.stream(st):
                    part.tick()

    def synchronize(self):
        self.torch.cuda.synchronize()

    def gather(self, name: str):
        """Concatenated per-robot tensor: a key of BatchEpisode.res ("status", "cmds", ...) or an attribute ("pose",
        "cmd_vel", "cmd_source", "proj_error")."""
        self.synchronize()
        return self.torch.cat([p.res[name] if name in p.res else getattr(p, name) for p in self.parts], dim=0)


def far_obstacle_grid(cells: int = 120, resolution: float = 0.1, origin=(-6.0, -6.0)):
    """An ObstacleDistance grid whose every cell points at one far corner obstacle (valid, but inert for the crowd)."""
    idx = np.zeros((cells, cells), np.uint32)  # all cells -> cell 0 (the grid corner)
    return idx, np.asarray(origin, np.float64), float(np.float32(resolution))


def arc_plans(pose0: np.ndarray, curvature: np.ndarray, L: int = 400, ds: float = 0.05):
    """Synthetic global plans: constant-curvature arcs of L poses (spacing ds) from each robot's start pose.
    Returns (plan [B,L,2], plan_len [B])."""
    B = pose0.shape[0]
    plan = np.zeros((B, L, 2))
    x, y, th = pose0[:, 0].copy(), pose0[:, 1].copy(), pose0[:, 2].copy()
    for i in range(L):
        plan[:, i, 0], plan[:, i, 1] = x, y
        x, y, th = x + ds * np.cos(th), y + ds * np.sin(th), th + curvature * ds
    return plan, np.full(B, L, np.int32)
