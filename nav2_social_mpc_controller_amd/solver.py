"""Python host side above the C ABI (include/smpc.h): loads csrc/libsmpc_hip.so and mirrors the part of the
reference's `Optimizer` interface that sits on the hot path (optimizer.hpp:152,167-170):

    Optimizer.initialize(OptimizerParams)   -> BatchSolver(params)
    Optimizer.optimize(...) for B scenes     -> BatchSolver.solve(scenes) / solve_device(...)

There is NO CPU fallback: a missing library or a missing HIP device raises.
"""
import ctypes as C
import os

import numpy as np

from . import _abi
from ._abi import (SmpcEvalOut, SmpcFormatBatch, SmpcFormatOut, SmpcMemoryBatch, SmpcParams, SmpcPeopleBatch, SmpcPlanWindowBatch,
                   SmpcProjectionBatch, SmpcResultBatch, SmpcSceneBatch, SmpcTrajectorizeBatch, SmpcTrajectorizeOut)
from .params import OptimizerParams, TrajectorizerParams
from .scenes import SceneBatch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SMPC_LIB_PATH", os.path.join(_HERE, "csrc", "libsmpc_hip.so"))  # env override: A/B builds
_lib = None


class SmpcError(RuntimeError):
    pass


def load_library():
    """dlopen libsmpc_hip.so (built by __graft_entry__.build() / csrc/build.sh). Fails loudly if absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise SmpcError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                        "(hipcc --offload-arch=gfx950). There is no CPU fallback for the solver.")
    # One HIP runtime per process: PyTorch-ROCm ships its own libamdhip64 and no longer finds a GPU when it is imported
    # after this library has initialised /opt/rocm's copy, so when torch is installed it goes first (it is only the
    # allocator / stream plumbing of the device-resident paths; the host-array paths never touch it).
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(LIB_PATH)
    lib.smpc_abi_version.restype = C.c_int
    lib.smpc_last_error.restype = C.c_char_p
    lib.smpc_params_default.argtypes = [C.POINTER(SmpcParams)]
    lib.smpc_params_default.restype = None
    lib.smpc_dims.restype = C.c_int
    lib.smpc_dims.argtypes = [C.POINTER(SmpcParams), C.c_int, C.c_int] + [C.POINTER(C.c_int)] * 6
    lib.smpc_create.restype = C.c_void_p
    lib.smpc_create.argtypes = [C.POINTER(SmpcParams), C.c_int]
    lib.smpc_destroy.argtypes = [C.c_void_p]
    lib.smpc_destroy.restype = None
    lib.smpc_set_stream.argtypes = [C.c_void_p, C.c_void_p]
    lib.smpc_set_stream.restype = C.c_int
    lib.smpc_set_solve_share.argtypes = [C.c_void_p, C.c_int32]
    lib.smpc_set_solve_share.restype = C.c_int
    lib.smpc_solve_slot_width.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32]
    lib.smpc_solve_slot_width.restype = C.c_int
    lib.smpc_solve_batch.argtypes = [C.c_void_p, C.POINTER(SmpcSceneBatch), C.POINTER(SmpcResultBatch)]
    lib.smpc_solve_batch.restype = C.c_int
    lib.smpc_eval_batch.argtypes = [C.c_void_p, C.POINTER(SmpcSceneBatch), C.c_void_p, C.POINTER(SmpcEvalOut)]
    lib.smpc_eval_batch.restype = C.c_int
    lib.smpc_project_people_batch.argtypes = [C.c_void_p, C.POINTER(SmpcProjectionBatch), C.c_void_p, C.c_void_p]
    lib.smpc_project_people_batch.restype = C.c_int
    lib.smpc_people_to_status_batch.argtypes = [C.c_void_p, C.POINTER(SmpcPeopleBatch), C.c_void_p, C.c_void_p]
    lib.smpc_people_to_status_batch.restype = C.c_int
    lib.smpc_format_to_optimize_batch.argtypes = [C.c_void_p, C.POINTER(SmpcFormatBatch), C.POINTER(SmpcFormatOut)]
    lib.smpc_format_to_optimize_batch.restype = C.c_int
    lib.smpc_memory_store_batch.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
                                            C.POINTER(SmpcMemoryBatch), C.c_void_p]
    lib.smpc_memory_store_batch.restype = C.c_int
    lib.smpc_trajectorize_path_batch.argtypes = [C.c_void_p, C.POINTER(SmpcTrajectorizeBatch), C.POINTER(SmpcTrajectorizeOut)]
    lib.smpc_trajectorize_path_batch.restype = C.c_int
    lib.smpc_transform_global_plan_batch.argtypes = [C.c_void_p, C.POINTER(SmpcPlanWindowBatch), C.c_void_p, C.c_void_p, C.c_void_p]
    lib.smpc_transform_global_plan_batch.restype = C.c_int
    lib.smpc_select_command_batch.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32] + [C.c_void_p] * 7
    lib.smpc_select_command_batch.restype = C.c_int
    lib.smpc_stage_people_batch.argtypes = [C.c_void_p, C.POINTER(SmpcSceneBatch), C.c_void_p, C.c_void_p]
    lib.smpc_stage_people_batch.restype = C.c_int
    lib.smpc_fp64_peak_probe.argtypes = [C.c_void_p, C.c_int32]
    lib.smpc_fp64_peak_probe.restype = C.c_double
    lib.smpc_math_probe.argtypes = [C.c_void_p, C.c_int32, C.c_int32] + [C.c_void_p] * 4
    lib.smpc_math_probe.restype = C.c_int
    lib.smpc_last_kernel_ms.argtypes = [C.c_void_p]
    lib.smpc_last_kernel_ms.restype = C.c_double
    if lib.smpc_abi_version() != _abi.SMPC_ABI_VERSION:
        raise SmpcError("libsmpc_hip.so ABI version mismatch")
    _lib = lib
    return lib


def _check(lib, rc, what):
    if rc != 0:
        raise SmpcError(f"{what} failed ({rc}): {lib.smpc_last_error().decode()}")


class BatchSolver:
    """One solver bound to one HIP device; mirrors Optimizer::initialize + Optimizer::optimize for B scenes."""

    def __init__(self, params: OptimizerParams, device: int = 0):
        self.lib = load_library()
        self.params = params
        self._cparams = params.to_c()
        self._h = self.lib.smpc_create(C.byref(self._cparams), int(device))
        if not self._h:
            raise SmpcError(f"smpc_create failed: {self.lib.smpc_last_error().decode()}")
        self.device = device

    def close(self):
        if getattr(self, "_h", None):
            self.lib.smpc_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_stream(self, stream_ptr: int):
        _check(self.lib, self.lib.smpc_set_stream(self._h, C.c_void_p(stream_ptr)), "smpc_set_stream")

    def set_solve_share(self, n: int):
        """smpc_set_solve_share: this handle's solve launches leave room for n - 1 concurrent ones (other streams)."""
        _check(self.lib, self.lib.smpc_set_solve_share(self._h, int(n)), "smpc_set_solve_share")

    def solve_slot_width(self, B: int, T: int, N: int) -> int:
        """smpc_solve_slot_width: 32 (two scenes per wave) or 64 (one) for a solve launch of this shape."""
        w = self.lib.smpc_solve_slot_width(self._h, int(B), int(T), int(N))
        if w < 0:
            _check(self.lib, w, "smpc_solve_slot_width")
        return w

    def math_probe(self, fn: int, a: np.ndarray, b: np.ndarray = None):
        """smpc_math_probe: the sweep's elementary functions evaluated on the device (see include/smpc.h)."""
        a = np.ascontiguousarray(a, dtype=np.float64)
        n = a.size // 8 if fn == 7 else a.size
        bb = None if b is None else np.ascontiguousarray(b, dtype=np.float64)
        o0 = np.empty(n)
        o1 = np.empty(n)
        _check(self.lib, self.lib.smpc_math_probe(self._h, int(fn), n, a.ctypes.data, None if bb is None else bb.ctypes.data,
                                                  o0.ctypes.data, o1.ctypes.data), "smpc_math_probe")
        return o0, o1

    def fp64_peak_tflops(self, iters: int = 20000) -> float:
        """Measured FP64 vector peak of this device (smpc_fp64_peak_probe)."""
        return float(self.lib.smpc_fp64_peak_probe(self._h, int(iters)))

    def last_kernel_ms(self) -> float:
        return float(self.lib.smpc_last_kernel_ms(self._h))

    # -- host-memory path (stages through HBM inside the library) ---------------------------------
    def solve(self, scenes: SceneBatch, order: np.ndarray = None):
        """Solve B scenes (host arrays in, host arrays out). order: optional queue order of the persistent kernel, a
        permutation of 0..B-1 (smpc_scene_batch.order: longest scenes first shortens a lone launch; results do not
        depend on it)."""
        CH, bl, nb, P, M, _ = self.params.dims(scenes.T, True)
        scenes.validate(P)
        B, T = scenes.B, scenes.T
        out = {
            "params": np.zeros((B, P)), "cmds": np.zeros((B, T + 1, 2)), "path": np.zeros((B, T + 1, 3)),
            "status": np.zeros(B, np.int32), "reason": np.zeros(B, np.int32), "iterations": np.zeros(B, np.int32),
            "evaluations": np.zeros(B, np.int32), "initial_cost": np.zeros(B), "final_cost": np.zeros(B),
        }
        rb = SmpcResultBatch()
        for k, v in out.items():
            setattr(rb, k, v.ctypes.data)
        sb = scenes.to_c()
        if order is not None:
            order = np.ascontiguousarray(order, np.int32)
            assert order.shape == (B,)
            sb.order = order.ctypes.data
        _check(self.lib, self.lib.smpc_solve_batch(self._h, C.byref(sb), C.byref(rb)), "smpc_solve_batch")
        return out

    @staticmethod
    def longest_first(evaluations):
        """Queue order for the next solve of the same (or the next control period's) scenes from the sweep counts of the
        last one: a torch int32 tensor on the device of `evaluations`, scenes with the most sweeps first."""
        import torch

        return torch.argsort(evaluations, descending=True, stable=True).to(torch.int32)

    def row_permutation(self, T: int, has_people: bool = True):
        """perm with reference_rows = critic_major_rows[perm] (smpc_eval_batch_out.row_order 1 -> 0) for one scene."""
        CH, bl, nb, P, M, _ = self.params.dims(T, has_people)
        rps = 8 if has_people else 5
        nfeas = M - rps * T
        perm = np.empty(M, np.int64)
        for t in range(T):
            base = rps * t + min(max(t - 1, 0), nfeas)
            for c in range(rps):
                perm[base + c] = c * T + t
            if 1 <= t <= nfeas:
                perm[base + rps] = rps * T + (t - 1)
        return perm

    def evaluate(self, scenes: SceneBatch, x: np.ndarray, row_order: int = 0):
        CH, bl, nb, P, M, _ = self.params.dims(scenes.T, True)
        scenes.validate(P)
        B = scenes.B
        x = np.ascontiguousarray(x, dtype=np.float64)
        assert x.shape == (B, P)
        out = {"residuals": np.zeros((B, M)), "jacobian": np.zeros((B, M, P)), "cost": np.zeros(B),
               "gradient": np.zeros((B, P))}
        eo = SmpcEvalOut()
        for k, v in out.items():
            setattr(eo, k, v.ctypes.data)
        eo.row_order = int(row_order)
        sb = scenes.to_c()
        _check(self.lib, self.lib.smpc_eval_batch(self._h, C.byref(sb), x.ctypes.data, C.byref(eo)), "smpc_eval_batch")
        return out

    # -- people projection: Optimizer::project_people for B scenes -------------
    def project_people(self, init_people: np.ndarray, robot_path: np.ndarray, od_indexes: np.ndarray,
                       od_origin: np.ndarray, od_resolution: float, max_time: float, time_step: float):
        """init_people [B,N,6], robot_path [B,T+1,6], od_indexes [B or 1,h,w] uint32, od_origin [B or 1,2].
        Returns (people_proj [B,T+1,6,N], error [B])."""
        init_people = np.ascontiguousarray(init_people, np.float64)
        robot_path = np.ascontiguousarray(robot_path, np.float64)
        od_indexes = np.ascontiguousarray(od_indexes, np.uint32)
        od_origin = np.ascontiguousarray(od_origin, np.float64)
        B, N, _ = init_people.shape
        T = robot_path.shape[1] - 1
        pb = SmpcProjectionBatch()
        pb.B, pb.T, pb.N, pb.on_device = B, T, N, 0
        pb.max_time, pb.time_step = float(max_time), float(time_step)
        pb.init_people, pb.robot_path = init_people.ctypes.data, robot_path.ctypes.data
        pb.od_indexes = od_indexes.ctypes.data if od_indexes.size else None
        pb.od_shared = 1 if od_indexes.shape[0] == 1 else 0
        pb.od_height, pb.od_width = (int(od_indexes.shape[1]), int(od_indexes.shape[2])) if od_indexes.ndim == 3 else (0, 0)
        pb.od_resolution = float(od_resolution)
        pb.od_origin = od_origin.ctypes.data
        out = np.zeros((B, T + 1, 6, N))
        err = np.zeros(B, np.int32)
        _check(self.lib, self.lib.smpc_project_people_batch(self._h, C.byref(pb), out.ctypes.data, err.ctypes.data),
               "smpc_project_people_batch")
        return out, err

    # -- initial-guess generator: PathTrajectorizer::trajectorize for B plans ------------------
    @staticmethod
    def trajectorize_c(tp: TrajectorizerParams, B: int, L: int, on_device: int) -> SmpcTrajectorizeBatch:
        tb = SmpcTrajectorizeBatch()
        tb.B, tb.L, tb.max_steps, tb.on_device = B, L, tp.max_steps, on_device
        tb.omnidirectional = 1 if tp.omnidirectional else 0
        tb.desired_linear_vel, tb.lookahead_dist = tp.desired_linear_vel, tp.lookahead_dist
        tb.max_angular_vel, tb.time_step = tp.max_angular_vel, tp.time_step
        return tb

    def trajectorize(self, tp: TrajectorizerParams, plan: np.ndarray, plan_len: np.ndarray, robot_pose: np.ndarray):
        """plan [B,L,2], plan_len [B], robot_pose [B,3] (x, y, yaw). Returns dict(path [B,S+1,3], cmds [B,S+1,2],
        cmds_vy [B,S+1], n_poses [B], error [B]) with S = tp.max_steps."""
        plan = np.ascontiguousarray(plan, np.float64)
        plan_len = np.ascontiguousarray(plan_len, np.int32)
        robot_pose = np.ascontiguousarray(robot_pose, np.float64)
        B, L, _ = plan.shape
        tb = self.trajectorize_c(tp, B, L, 0)
        tb.plan, tb.plan_len, tb.robot_pose = plan.ctypes.data, plan_len.ctypes.data, robot_pose.ctypes.data
        S1 = tp.max_steps + 1
        out = {"path": np.zeros((B, S1, 3)), "cmds": np.zeros((B, S1, 2)), "cmds_vy": np.zeros((B, S1)),
               "n_poses": np.zeros(B, np.int32), "error": np.zeros(B, np.int32)}
        to = SmpcTrajectorizeOut()
        for k, v in out.items():
            setattr(to, k, v.ctypes.data)
        _check(self.lib, self.lib.smpc_trajectorize_path_batch(self._h, C.byref(tb), C.byref(to)), "smpc_trajectorize_path_batch")
        return out

    def transform_global_plan(self, plan: np.ndarray, plan_len: np.ndarray, plan_start: np.ndarray, robot_pose: np.ndarray,
                              max_robot_pose_search_dist: float, dist_threshold: float, to_local: np.ndarray = None):
        """PathHandler::transformGlobalPlan for B robots (smpc_transform_global_plan_batch): plan [B,L,2], plan_len [B],
        plan_start [B] (updated in place: the pruning), robot_pose [B,3] in the plan frame, to_local [B,3] or None.
        Returns dict(window [B,L,2], window_len [B], error [B])."""
        plan = np.ascontiguousarray(plan, np.float64)
        plan_len = np.ascontiguousarray(plan_len, np.int32)
        assert plan_start.dtype == np.int32 and plan_start.flags.c_contiguous
        robot_pose = np.ascontiguousarray(robot_pose, np.float64)
        B, L, _ = plan.shape
        wb = SmpcPlanWindowBatch()
        wb.B, wb.L, wb.on_device = B, L, 0
        wb.max_robot_pose_search_dist, wb.dist_threshold = float(max_robot_pose_search_dist), float(dist_threshold)
        wb.plan, wb.plan_len, wb.plan_start, wb.robot_pose = plan.ctypes.data, plan_len.ctypes.data, plan_start.ctypes.data, robot_pose.ctypes.data
        if to_local is not None:
            to_local = np.ascontiguousarray(to_local, np.float64)
            wb.to_local = to_local.ctypes.data
        out = {"window": np.zeros((B, L, 2)), "window_len": np.zeros(B, np.int32), "error": np.zeros(B, np.int32)}
        _check(self.lib, self.lib.smpc_transform_global_plan_batch(self._h, C.byref(wb), out["window"].ctypes.data,
                                                                   out["window_len"].ctypes.data, out["error"].ctypes.data),
               "smpc_transform_global_plan_batch")
        return out

    def transform_global_plan_device(self, wb: "SmpcPlanWindowBatch", window_ptr: int, window_len_ptr: int, error_ptr: int = 0):
        assert wb.on_device == 1
        _check(self.lib, self.lib.smpc_transform_global_plan_batch(self._h, C.byref(wb), C.c_void_p(window_ptr),
                                                                   C.c_void_p(window_len_ptr), C.c_void_p(error_ptr)),
               "smpc_transform_global_plan_batch")

    def trajectorize_device(self, tb: SmpcTrajectorizeBatch, to: SmpcTrajectorizeOut):
        assert tb.on_device == 1
        _check(self.lib, self.lib.smpc_trajectorize_path_batch(self._h, C.byref(tb), C.byref(to)), "smpc_trajectorize_path_batch")

    def people_to_status(self, people: np.ndarray, count: np.ndarray, N: int = 3, robot_pose: np.ndarray = None,
                         fov_angle: float = np.pi / 4, costmap_origin: np.ndarray = None, size_x: int = 0, size_y: int = 0,
                         resolution: float = 0.0):
        """Optimizer::people_to_status for B scenes: people [B,Np,5] (px, py, vx, vy, vz), count [B] ->
        (init_people [B,N,6], has_people [B] uint8). With robot_pose [B,3] the field-of-view filter of
        computeVelocityCommands runs first (costmap_origin [B or 1,2], size_x/size_y cells, resolution)."""
        people = np.ascontiguousarray(people, np.float64)
        count = np.ascontiguousarray(count, np.int32)
        B, Np, _ = people.shape
        pb = SmpcPeopleBatch()
        pb.B, pb.Np, pb.N, pb.on_device = B, Np, N, 0
        pb.people, pb.count = people.ctypes.data, count.ctypes.data
        if robot_pose is not None:
            robot_pose = np.ascontiguousarray(robot_pose, np.float64)
            costmap_origin = np.ascontiguousarray(costmap_origin, np.float64).reshape(-1, 2)
            pb.robot_pose, pb.fov_angle = robot_pose.ctypes.data, float(fov_angle)
            pb.costmap_origin, pb.costmap_shared = costmap_origin.ctypes.data, 1 if costmap_origin.shape[0] == 1 else 0
            pb.size_x, pb.size_y, pb.resolution = int(size_x), int(size_y), float(resolution)
        out, has = np.zeros((B, N, 6)), np.zeros(B, np.uint8)
        _check(self.lib, self.lib.smpc_people_to_status_batch(self._h, C.byref(pb), out.ctypes.data, has.ctypes.data),
               "smpc_people_to_status_batch")
        return out, has

    def select_command(self, traj_n_poses, traj_cmds: np.ndarray, status: np.ndarray, cmds: np.ndarray, window_error=None):
        """The command computeVelocityCommands returns for B robots (fallbacks included): traj_cmds [B,rows,2],
        status [B], cmds [B,T+1,2], traj_n_poses [B] or None, window_error [B] (smpc_window_error of this cycle) or None.
        Returns (cmd_vel [B,2], source [B])."""
        traj_cmds = np.ascontiguousarray(traj_cmds, np.float64)
        status = np.ascontiguousarray(status, np.int32)
        cmds = np.ascontiguousarray(cmds, np.float64)
        B, rows, _ = traj_cmds.shape
        T = cmds.shape[1] - 1
        n = None if traj_n_poses is None else np.ascontiguousarray(traj_n_poses, np.int32)
        out, src = np.zeros((B, 2)), np.zeros(B, np.int32)
        we = None if window_error is None else np.ascontiguousarray(window_error, np.int32)
        _check(self.lib, self.lib.smpc_select_command_batch(self._h, B, T, rows, 0, None if n is None else n.ctypes.data,
                                                            traj_cmds.ctypes.data, status.ctypes.data, cmds.ctypes.data,
                                                            out.ctypes.data, src.ctypes.data,
                                                            None if we is None else we.ctypes.data), "smpc_select_command_batch")
        return out, src

    def select_command_device(self, B, T, rows, traj_n_ptr, traj_cmds_ptr, status_ptr, cmds_ptr, cmd_vel_ptr, source_ptr,
                              window_error_ptr=0):
        _check(self.lib, self.lib.smpc_select_command_batch(self._h, B, T, rows, 1, C.c_void_p(traj_n_ptr), C.c_void_p(traj_cmds_ptr),
                                                            C.c_void_p(status_ptr), C.c_void_p(cmds_ptr), C.c_void_p(cmd_vel_ptr),
                                                            C.c_void_p(source_ptr), C.c_void_p(window_error_ptr)),
               "smpc_select_command_batch")

    # -- warm start / input formatting: format_to_optimize + TrajectoryMemory for B scenes -----
    def format_to_optimize(self, path: np.ndarray, cmds: np.ndarray, speed: np.ndarray, memory: dict,
                           current_path_w: float = None, current_cmds_w: float = None, n_poses=None, max_poses: int = 0,
                           T: int = None):
        """path [B,rows,3] (x, y, yaw), cmds [B,rows,2], speed [B,2]; memory = new_memory(B, T) (updated in place when a
        record is empty). T: horizon (stride) of the outputs, default rows - 1. n_poses [B] + max_poses: horizons per
        scene (smpc_format_batch.n_poses; the memory then needs its `length` array: new_memory(..., lengths=True)).
        Returns dict(robot_status [B,T+1,6], pose0, init_params, path_pts, goal_yaw, T_scene)."""
        path = np.ascontiguousarray(path, np.float64)
        cmds = np.ascontiguousarray(cmds, np.float64)
        speed = np.ascontiguousarray(speed, np.float64)
        B, rows, _ = path.shape
        T = rows - 1 if T is None else int(T)
        Tp = T + 1
        assert cmds.shape == (B, rows, 2) and speed.shape == (B, 2) and rows >= Tp
        assert memory["prev_path"].shape == (B, Tp, 3) and memory["prev_cmds"].shape == (B, Tp, 2)
        CH, bl, nb, P, M, _ = self.params.dims(T, True)
        fb = SmpcFormatBatch()
        fb.B, fb.T, fb.path_rows, fb.on_device = B, T, rows, 0
        if n_poses is not None:
            n_poses = np.ascontiguousarray(n_poses, np.int32)
            assert n_poses.shape == (B,) and "length" in memory
            fb.n_poses, fb.max_poses = n_poses.ctypes.data, int(max_poses)
        if "length" in memory:
            fb.memory.length = memory["length"].ctypes.data
        fb.time_step = float(self.params.dt)
        fb.current_path_w = float(self.params.current_path_weight if current_path_w is None else current_path_w)
        fb.current_cmds_w = float(self.params.current_cmds_weight if current_cmds_w is None else current_cmds_w)
        fb.path, fb.cmds, fb.speed = path.ctypes.data, cmds.ctypes.data, speed.ctypes.data
        fb.memory.prev_path = memory["prev_path"].ctypes.data
        fb.memory.prev_cmds = memory["prev_cmds"].ctypes.data
        fb.memory.valid = memory["valid"].ctypes.data
        out = {"robot_status": np.zeros((B, Tp, 6)), "pose0": np.zeros((B, 3)), "init_params": np.zeros((B, P)),
               "path_pts": np.zeros((B, Tp, 2)), "goal_yaw": np.zeros(B), "T_scene": np.zeros(B, np.int32)}
        fo = SmpcFormatOut()
        for k, v in out.items():
            setattr(fo, k, v.ctypes.data)
        _check(self.lib, self.lib.smpc_format_to_optimize_batch(self._h, C.byref(fb), C.byref(fo)),
               "smpc_format_to_optimize_batch")
        return out

    @staticmethod
    def new_memory(B: int, T: int, lengths: bool = False):
        """An empty TrajectoryMemory record per scene (host arrays). lengths: with the per-record sizes that scenes with
        horizons of their own need (smpc_memory_batch.length)."""
        m = {"prev_path": np.zeros((B, T + 1, 3)), "prev_cmds": np.zeros((B, T + 1, 2)), "valid": np.zeros(B, np.int32)}
        if lengths:
            m["length"] = np.zeros((B, 2), np.int32)
        return m

    def memory_store(self, status: np.ndarray, path: np.ndarray, cmds: np.ndarray, memory: dict, T_scene=None):
        """The store at the end of Optimizer::optimize: usable solves (status != FAILURE) overwrite their record (the
        first T_scene[b] + 1 rows when the scenes have horizons of their own)."""
        status = np.ascontiguousarray(status, np.int32)
        path = np.ascontiguousarray(path, np.float64)
        cmds = np.ascontiguousarray(cmds, np.float64)
        B, Tp, _ = path.shape
        mb = SmpcMemoryBatch()
        mb.prev_path, mb.prev_cmds, mb.valid = (memory["prev_path"].ctypes.data, memory["prev_cmds"].ctypes.data,
                                                memory["valid"].ctypes.data)
        if "length" in memory:
            mb.length = memory["length"].ctypes.data
        ts = None if T_scene is None else np.ascontiguousarray(T_scene, np.int32)
        _check(self.lib, self.lib.smpc_memory_store_batch(self._h, B, Tp - 1, 0, status.ctypes.data, path.ctypes.data,
                                                          cmds.ctypes.data, C.byref(mb), None if ts is None else ts.ctypes.data),
               "smpc_memory_store_batch")

    def format_device(self, fb: SmpcFormatBatch, fo: SmpcFormatOut):
        assert fb.on_device == 1
        _check(self.lib, self.lib.smpc_format_to_optimize_batch(self._h, C.byref(fb), C.byref(fo)),
               "smpc_format_to_optimize_batch")

    def memory_store_device(self, B: int, T: int, status_ptr: int, path_ptr: int, cmds_ptr: int, mb: SmpcMemoryBatch,
                            T_scene_ptr: int = 0):
        _check(self.lib, self.lib.smpc_memory_store_batch(self._h, B, T, 1, C.c_void_p(status_ptr), C.c_void_p(path_ptr),
                                                          C.c_void_p(cmds_ptr), C.byref(mb), C.c_void_p(T_scene_ptr)),
               "smpc_memory_store_batch")

    def people_to_status_device(self, pb: SmpcPeopleBatch, out_ptr: int, has_people_ptr: int):
        assert pb.on_device == 1
        _check(self.lib, self.lib.smpc_people_to_status_batch(self._h, C.byref(pb), C.c_void_p(out_ptr), C.c_void_p(has_people_ptr)),
               "smpc_people_to_status_batch")

    def project_people_device(self, pb: SmpcProjectionBatch, out_ptr: int, err_ptr: int):
        assert pb.on_device == 1
        _check(self.lib, self.lib.smpc_project_people_batch(self._h, C.byref(pb), C.c_void_p(out_ptr), C.c_void_p(err_ptr)),
               "smpc_project_people_batch")

    # -- device-resident path (inputs already in HBM; asynchronous on the handle's stream) -------
    def alloc_results(self, B: int, T: int, device="cuda:0"):
        import torch

        CH, bl, nb, P, M, _ = self.params.dims(T, True)
        t = {
            "params": torch.empty((B, P), dtype=torch.float64, device=device),
            "cmds": torch.empty((B, T + 1, 2), dtype=torch.float64, device=device),
            "path": torch.empty((B, T + 1, 3), dtype=torch.float64, device=device),
            "status": torch.empty(B, dtype=torch.int32, device=device),
            "reason": torch.empty(B, dtype=torch.int32, device=device),
            "iterations": torch.empty(B, dtype=torch.int32, device=device),
            "evaluations": torch.empty(B, dtype=torch.int32, device=device),
            "initial_cost": torch.empty(B, dtype=torch.float64, device=device),
            "final_cost": torch.empty(B, dtype=torch.float64, device=device),
        }
        rb = SmpcResultBatch()
        for k, v in t.items():
            setattr(rb, k, v.data_ptr())
        return rb, t

    def stage_people(self, scenes: SceneBatch):
        """smpc_stage_people_batch on host arrays: (records [B,N,T,4], aux [B,T,2])."""
        B, T, N = scenes.B, scenes.T, scenes.N
        rec = np.zeros((B, N, T, 4))
        aux = np.zeros((B, T, 2))
        sb = scenes.to_c()
        _check(self.lib, self.lib.smpc_stage_people_batch(self._h, C.byref(sb), rec.ctypes.data, aux.ctypes.data),
               "smpc_stage_people_batch")
        return rec, aux

    def stage_people_device(self, sb: SmpcSceneBatch, device="cuda:0"):
        """Stage the device-resident people block of `sb` once and attach the result to it: later solve_device /
        eval_device calls with this batch read the staged records instead of running the staging pass again.
        Returns the tensors that own the memory (keep them alive as long as `sb` is used)."""
        import torch

        assert sb.on_device == 1
        rec = torch.zeros((sb.B, sb.N, sb.T, 4), dtype=torch.float64, device=device)
        aux = torch.zeros((sb.B, sb.T, 2), dtype=torch.float64, device=device)
        _check(self.lib, self.lib.smpc_stage_people_batch(self._h, C.byref(sb), C.c_void_p(rec.data_ptr()),
                                                          C.c_void_p(aux.data_ptr())), "smpc_stage_people_batch")
        sb.people_records, sb.people_aux = rec.data_ptr(), aux.data_ptr()
        return rec, aux

    def solve_device(self, sb: SmpcSceneBatch, rb: SmpcResultBatch):
        assert sb.on_device == 1
        _check(self.lib, self.lib.smpc_solve_batch(self._h, C.byref(sb), C.byref(rb)), "smpc_solve_batch")

    def alloc_eval(self, B: int, T: int, device="cuda:0", row_order: int = 0):
        import torch

        CH, bl, nb, P, M, _ = self.params.dims(T, True)
        t = {
            "residuals": torch.empty((B, M), dtype=torch.float64, device=device),
            "jacobian": torch.empty((B, M, P), dtype=torch.float64, device=device),
            "cost": torch.empty(B, dtype=torch.float64, device=device),
            "gradient": torch.empty((B, P), dtype=torch.float64, device=device),
        }
        eo = SmpcEvalOut()
        for k, v in t.items():
            setattr(eo, k, v.data_ptr())
        eo.row_order = int(row_order)
        return eo, t

    def eval_device(self, sb: SmpcSceneBatch, x_ptr: int, eo: SmpcEvalOut):
        assert sb.on_device == 1
        _check(self.lib, self.lib.smpc_eval_batch(self._h, C.byref(sb), C.c_void_p(x_ptr), C.byref(eo)), "smpc_eval_batch")
