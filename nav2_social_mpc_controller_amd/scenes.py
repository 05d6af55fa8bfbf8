"""Scene batches for the batched MPC solver: the inputs `Optimizer::optimize` hands to the Ceres problem
(reference src/optimizer.cpp:197-237) in structure-of-arrays form, plus the seeded synthetic crowd-scene
generator SURVEY.md §8(d) specifies (counter-based SplitMix64 keyed by (seed, scene_id, field_id), so every
shard / the CPU oracle regenerate identical scenes with no communication).
"""
import ctypes as C
from dataclasses import dataclass
from typing import Optional

import numpy as np

from ._abi import SmpcSceneBatch
from .params import OptimizerParams

_MASK = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        z = (x + np.uint64(0x9E3779B97F4A7C15)) & _MASK
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _MASK
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _MASK
        return z ^ (z >> np.uint64(31))


def uniform(seed: int, scene_ids: np.ndarray, field_id: int, n: int = 1) -> np.ndarray:
    """U[0,1) doubles of shape [len(scene_ids), n], a pure function of (seed, scene_id, field_id, k)."""
    sid = scene_ids.astype(np.uint64)[:, None]
    k = np.arange(n, dtype=np.uint64)[None, :]
    with np.errstate(over="ignore"):
        key = _splitmix64(np.uint64(seed) ^ _splitmix64(sid * np.uint64(0x100000001B3) + np.uint64(field_id)))
        bits = _splitmix64(key + k * np.uint64(0xD1342543DE82EF95))
    return (bits >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


@dataclass
class SceneBatch:
    """SoA inputs of B independent MPC problems (layout = include/smpc.h `smpc_scene_batch`)."""
    T: int
    N: int
    dt: float
    pose0: np.ndarray          # [B,3]
    init_params: np.ndarray    # [B,P]
    path_pts: np.ndarray       # [B,T+1,2]
    goal_yaw: np.ndarray       # [B]
    people: np.ndarray         # [B,T+1,6,N]
    has_people: np.ndarray     # [B] uint8
    costmap: np.ndarray        # [B or 1,size_y,size_x] uint8
    costmap_origin: np.ndarray  # [B or 1,2]
    resolution: float
    costmap_shared: bool = False
    # optional horizon of each scene, 1 <= T_scene[b] <= T (smpc_scene_batch.T_scene): of path_pts / people the first
    # T_scene[b] + 1 rows of scene b count, of init_params the first P_b entries; None: every scene has T steps
    T_scene: Optional[np.ndarray] = None

    @property
    def B(self) -> int:
        return int(self.pose0.shape[0])

    @property
    def size_y(self) -> int:
        return int(self.costmap.shape[1])

    @property
    def size_x(self) -> int:
        return int(self.costmap.shape[2])

    def validate(self, P: int):
        B, T, N = self.B, self.T, self.N
        assert self.pose0.shape == (B, 3) and self.pose0.dtype == np.float64
        assert self.init_params.shape == (B, P) and self.init_params.dtype == np.float64
        assert self.path_pts.shape == (B, T + 1, 2) and self.path_pts.dtype == np.float64
        assert self.goal_yaw.shape == (B,) and self.goal_yaw.dtype == np.float64
        assert self.people.shape == (B, T + 1, 6, N) and self.people.dtype == np.float64
        assert self.has_people.shape == (B,) and self.has_people.dtype == np.uint8
        nb_maps = 1 if self.costmap_shared else B
        assert self.costmap.shape[0] == nb_maps and self.costmap.dtype == np.uint8
        assert self.costmap_origin.shape == (nb_maps, 2) and self.costmap_origin.dtype == np.float64
        for a in (self.pose0, self.init_params, self.path_pts, self.goal_yaw, self.people, self.has_people,
                  self.costmap, self.costmap_origin):
            assert a.flags["C_CONTIGUOUS"]
        if self.T_scene is not None:
            assert self.T_scene.shape == (B,) and self.T_scene.dtype == np.int32 and self.T_scene.flags["C_CONTIGUOUS"]

    def to_c(self) -> SmpcSceneBatch:
        """C view over the host arrays (arrays stay owned by this object)."""
        sb = SmpcSceneBatch()
        sb.B, sb.T, sb.N, sb.on_device = self.B, self.T, self.N, 0
        sb.dt = self.dt
        sb.pose0 = self.pose0.ctypes.data
        sb.init_params = self.init_params.ctypes.data
        sb.path_pts = self.path_pts.ctypes.data
        sb.goal_yaw = self.goal_yaw.ctypes.data
        sb.people = self.people.ctypes.data
        sb.has_people = self.has_people.ctypes.data
        sb.costmap = self.costmap.ctypes.data
        sb.costmap_shared = 1 if self.costmap_shared else 0
        sb.size_x, sb.size_y = self.size_x, self.size_y
        sb.costmap_origin = self.costmap_origin.ctypes.data
        sb.resolution = self.resolution
        if self.T_scene is not None:
            sb.T_scene = self.T_scene.ctypes.data
        return sb

    def to_device(self, device="cuda:0"):
        """Copy to HBM as torch tensors; returns (SmpcSceneBatch with device pointers, dict of tensors)."""
        import torch

        t = {k: torch.from_numpy(getattr(self, k)).to(device) for k in
             ("pose0", "init_params", "path_pts", "goal_yaw", "people", "has_people", "costmap", "costmap_origin")}
        if self.T_scene is not None:
            t["T_scene"] = torch.from_numpy(self.T_scene).to(device)
        sb = SmpcSceneBatch()
        sb.B, sb.T, sb.N, sb.on_device = self.B, self.T, self.N, 1
        sb.dt = self.dt
        for k, v in t.items():
            setattr(sb, k, v.data_ptr())
        sb.costmap_shared = 1 if self.costmap_shared else 0
        sb.size_x, sb.size_y = self.size_x, self.size_y
        sb.resolution = self.resolution
        return sb, t

    def select(self, idx) -> "SceneBatch":
        idx = np.atleast_1d(np.asarray(idx))
        cm = self.costmap if self.costmap_shared else np.ascontiguousarray(self.costmap[idx])
        co = self.costmap_origin if self.costmap_shared else np.ascontiguousarray(self.costmap_origin[idx])
        return SceneBatch(self.T, self.N, self.dt, np.ascontiguousarray(self.pose0[idx]),
                          np.ascontiguousarray(self.init_params[idx]), np.ascontiguousarray(self.path_pts[idx]),
                          np.ascontiguousarray(self.goal_yaw[idx]), np.ascontiguousarray(self.people[idx]),
                          np.ascontiguousarray(self.has_people[idx]), cm, co, self.resolution, self.costmap_shared,
                          None if self.T_scene is None else np.ascontiguousarray(self.T_scene[idx]))

    def with_horizons(self, T_scene) -> "SceneBatch":
        """The same scenes with a horizon per scene: scene b keeps its first T_scene[b] + 1 poses / people rows; its goal
        heading becomes the heading the path has at its own last pose (scenes built by make_scenes: the reference path
        is an arc, so that is the heading after T_scene[b] steps)."""
        T_scene = np.ascontiguousarray(T_scene, np.int32)
        assert T_scene.shape == (self.B,) and T_scene.min() >= 1 and T_scene.max() <= self.T
        out = self.select(np.arange(self.B))
        out.T_scene = T_scene
        return out

    def cut(self, idx, Tb: int, P_b: int) -> "SceneBatch":
        """Scenes idx as a batch whose T is Tb: what a caller with exactly these horizons would hand over (arrays
        truncated to Tb + 1 rows and P_b parameters)."""
        sub = self.select(idx)
        return SceneBatch(Tb, self.N, self.dt, sub.pose0, np.ascontiguousarray(sub.init_params[:, :P_b]),
                          np.ascontiguousarray(sub.path_pts[:, :Tb + 1]), sub.goal_yaw,
                          np.ascontiguousarray(sub.people[:, :Tb + 1]), sub.has_people, sub.costmap, sub.costmap_origin,
                          self.resolution, self.costmap_shared)

    def save(self, path: str):
        np.savez_compressed(path, T=self.T, N=self.N, dt=self.dt, pose0=self.pose0, init_params=self.init_params,
                            path_pts=self.path_pts, goal_yaw=self.goal_yaw, people=self.people,
                            has_people=self.has_people, costmap=self.costmap, costmap_origin=self.costmap_origin,
                            resolution=self.resolution, costmap_shared=self.costmap_shared)

    @staticmethod
    def load(path: str) -> "SceneBatch":
        z = np.load(path, allow_pickle=False)
        return SceneBatch(int(z["T"]), int(z["N"]), float(z["dt"]), z["pose0"], z["init_params"], z["path_pts"],
                          z["goal_yaw"], z["people"], z["has_people"], z["costmap"], z["costmap_origin"],
                          float(z["resolution"]), bool(z["costmap_shared"]))


def make_scenes(params: OptimizerParams, B: int, N: int, seed: int = 0x5EED0001, first_scene: int = 0,
                T: Optional[int] = None, map_cells: int = 200, resolution: float = 0.05,
                standing_fraction: float = 0.2, people_present: bool = True,
                n_valid: Optional[int] = None) -> SceneBatch:
    """Synthetic crowd scenes per SURVEY.md §8(d).

    pose0 near the costmap centre, a constant-curvature reference path (mirrors the trajectorizer output),
    initial parameter blocks reproducing the aliasing quirk (src/optimizer.cpp:254-261: blocks start from the
    speeds at *time steps* 0,1,2,...), constant-velocity people, u8 costmap with inflated discs.
    `n_valid` (<= N) marks agents n_valid..N-1 invalid (t = -1 at the origin), like people_to_status pads.
    """
    if T is None:
        T = params.rollout_steps
    dt = params.dt
    CH, bl, nb, P, M, _ = params.dims(T, True)
    ids = np.arange(first_scene, first_scene + B, dtype=np.int64)
    u = lambda fid, n=1: uniform(seed, ids, fid, n)

    side = map_cells * resolution
    origin = (u(1, 2) - 0.5) * 10.0                      # costmap origin, a few metres from the world origin
    centre = origin + 0.5 * side
    pos0 = centre + (u(2, 2) - 0.5)                       # centre + U(-0.5,0.5)^2
    yaw0 = (u(3)[:, 0] * 2.0 - 1.0) * np.pi
    v_cur = u(4)[:, 0] * 0.6
    w_cur = u(5)[:, 0] - 0.5
    w_ref = (u(6)[:, 0] * 2.0 - 1.0) * 0.6
    pose0 = np.stack([pos0[:, 0], pos0[:, 1], yaw0], axis=1)

    # reference path: unicycle rollout with v = 0.6, w = w_ref, T+1 poses starting at pose0
    path = np.zeros((B, T + 1, 2))
    x, y, th = pos0[:, 0].copy(), pos0[:, 1].copy(), yaw0.copy()
    path[:, 0, 0], path[:, 0, 1] = x, y
    for k in range(1, T + 1):
        x = x + 0.6 * np.cos(th) * dt
        y = y + 0.6 * np.sin(th) * dt
        th = th + w_ref * dt
        path[:, k, 0], path[:, k, 1] = x, y
    goal_yaw = th - w_ref * dt  # yaw of the last pose (pose T has heading after T-1 turns)

    init = np.zeros((B, P))
    init[:, 0], init[:, 1] = v_cur, w_cur
    for b in range(1, nb):
        init[:, 2 * b], init[:, 2 * b + 1] = 0.6, w_ref

    people = np.zeros((B, T + 1, 6, max(N, 1)))
    if N > 0:
        r = 0.8 + u(10, N) * 2.7
        phi = (u(11, N) * 2.0 - 1.0) * np.pi
        heading = (u(12, N) * 2.0 - 1.0) * np.pi
        standing = u(13, N) < standing_fraction
        lv = np.where(standing, 0.0, 0.2 + u(14, N))
        px0 = pos0[:, 0:1] + r * np.cos(phi)
        py0 = pos0[:, 1:2] + r * np.sin(phi)
        for k in range(T + 1):
            people[:, k, 0, :] = px0 + lv * np.cos(heading) * (k * dt)
            people[:, k, 1, :] = py0 + lv * np.sin(heading) * (k * dt)
            people[:, k, 2, :] = heading
            people[:, k, 3, :] = k * dt
            people[:, k, 4, :] = lv
            people[:, k, 5, :] = 0.0
        if n_valid is not None and n_valid < N:
            people[:, :, :, n_valid:] = 0.0
            people[:, :, 3, n_valid:] = -1.0
    people = np.ascontiguousarray(people[:, :, :, :N]) if N > 0 else np.zeros((B, T + 1, 6, 0))
    has_people = np.full(B, 1 if (people_present and N > 0) else 0, dtype=np.uint8)

    # costmap: K ~ U{3..8} discs (cost 254) with exponential inflation 252*exp(-3 d) out to 0.7 m
    costmap = np.zeros((B, map_cells, map_cells), dtype=np.uint8)
    K = 3 + np.floor(u(20)[:, 0] * 6.0).astype(np.int64)
    win = int(np.ceil((0.5 + 0.7) / resolution)) + 1
    offs = np.arange(-win, win + 1)
    for k in range(8):
        cr = 0.15 + u(21 + 4 * k)[:, 0] * 0.35
        # disc centre: anywhere in the map but >= 1.0 m from the start position
        ang = u(22 + 4 * k)[:, 0] * 2.0 * np.pi
        dist = 1.0 + u(23 + 4 * k)[:, 0] * (0.5 * side - 1.2)
        cx = pos0[:, 0] + dist * np.cos(ang)
        cy = pos0[:, 1] + dist * np.sin(ang)
        active = k < K
        ci = np.floor((cx - origin[:, 0]) / resolution).astype(np.int64)
        cj = np.floor((cy - origin[:, 1]) / resolution).astype(np.int64)
        ii = ci[:, None] + offs[None, :]             # x cells [B,W]
        jj = cj[:, None] + offs[None, :]             # y cells [B,W]
        wx = origin[:, 0:1] + (ii + 0.5) * resolution
        wy = origin[:, 1:2] + (jj + 0.5) * resolution
        d = np.sqrt((wx[:, None, :] - cx[:, None, None]) ** 2 + (wy[:, :, None] - cy[:, None, None]) ** 2) - cr[:, None, None]
        cost = np.where(d <= 0.0, 254.0, np.where(d <= 0.7, np.floor(252.0 * np.exp(-3.0 * d)), 0.0))
        cost = np.where(active[:, None, None], cost, 0.0).astype(np.uint8)
        inb = (ii[:, None, :] >= 0) & (ii[:, None, :] < map_cells) & (jj[:, :, None] >= 0) & (jj[:, :, None] < map_cells)
        bi = np.broadcast_to(np.arange(B)[:, None, None], cost.shape)[inb]
        yi = np.broadcast_to(jj[:, :, None], cost.shape)[inb]
        xi = np.broadcast_to(ii[:, None, :], cost.shape)[inb]
        # (b, y, x) triples are unique within one disc, so a gather / max / scatter is exact
        costmap[bi, yi, xi] = np.maximum(costmap[bi, yi, xi], cost[inb])

    sb = SceneBatch(T, N, dt, np.ascontiguousarray(pose0), np.ascontiguousarray(init), np.ascontiguousarray(path),
                    np.ascontiguousarray(goal_yaw), people, has_people, costmap, np.ascontiguousarray(origin),
                    resolution, False)
    sb.validate(P)
    return sb
