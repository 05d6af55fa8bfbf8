"""Host-side mirror of the reference's `OptimizerParams` (optimizer.hpp:59-101, src/optimizer.cpp:16-85).

Field names follow the reference's ROS parameter names (`<plugin>.optimizer.*`, `<plugin>.optimizer.weights.*`,
`<plugin>.trajectorizer.{time_step,max_time}`); defaults are the reference's *code* defaults. `from_yaml`
reads a params/*.yaml-shaped file (the `FollowPath:` section of a Nav2 controller_server configuration).
"""
from dataclasses import dataclass, asdict

import numpy as np

from ._abi import LINEAR_SOLVER, SmpcParams


@dataclass
class TrajectorizerParams:
    """`<plugin>.trajectorizer.*` of PathTrajectorizer::configure (src/path_trajectorizer.cpp:53-84), code defaults."""
    omnidirectional: bool = False
    desired_linear_vel: float = 0.4
    lookahead_dist: float = 0.4
    max_angular_vel: float = 1.0
    time_step: float = 0.05
    max_time: float = 3.0

    @property
    def max_steps(self) -> int:
        """(int)round(max_time / time_step), doubles (src/path_trajectorizer.cpp:84)."""
        return int(np.round(self.max_time / self.time_step))


@dataclass
class OptimizerParams:
    # optimizer.* (src/optimizer.cpp:26-55, 76-83)
    linear_solver_type: str = "SPARSE_NORMAL_CHOLESKY"
    param_tol: float = 1e-15
    fn_tol: float = 1e-7
    gradient_tol: float = 1e-10
    max_iterations: int = 100
    debug_optimizer: bool = False
    control_horizon: int = 5
    parameter_block_length: int = 5
    current_path_weight: float = 1.0
    current_cmds_weight: float = 1.0
    # optimizer.weights.* (src/optimizer.cpp:57-75)
    distance_weight: float = 3.0
    social_weight: float = 1.0
    velocity_weight: float = 0.5
    angle_weight: float = 0.0
    agent_angle_weight: float = 0.5
    proxemics_weight: float = 90.0
    velocity_feasibility_weight: float = 0.5
    obstacle_weight: float = 0.0
    goal_align_weight: float = 0.0
    # trajectorizer.* read by the optimiser path (src/optimizer.cpp:84, src/path_trajectorizer.cpp:58-59)
    time_step: float = 0.05
    max_time: float = 3.0
    # literals of Optimizer::optimize (src/optimizer.cpp:238, 375-378)
    desired_linear_vel: float = 0.6
    v_min: float = 0.0
    v_max: float = 0.6
    w_min: float = -1.4
    w_max: float = 1.4
    # switches that have no reference counterpart (0 = reference behaviour)
    fixed_iterations: int = 0
    tol_needs_successful_step: int = 0

    def __post_init__(self):
        if self.linear_solver_type not in LINEAR_SOLVER:
            # same error behaviour as src/optimizer.cpp:31-45
            raise RuntimeError("Invalid parameter: linear_solver_type")

    @property
    def dt(self) -> float:
        """time_step as Optimizer::optimize sees it: a float widened to double (optimizer.hpp:170)."""
        return float(np.float32(self.time_step))

    @property
    def rollout_steps(self) -> int:
        """T for a trajectorized path longer than max_time: format_to_optimize cuts to round(max_time/dt)-1
        poses (src/optimizer.cpp:492-497) and optimize pops one velocity (:237)."""
        maxsize = int(np.round(np.float32(self.max_time) / np.float32(self.time_step)))
        return maxsize - 2

    def to_c(self) -> SmpcParams:
        p = SmpcParams()
        p.distance_w = self.distance_weight
        p.socialwork_w = self.social_weight
        p.velocity_w = self.velocity_weight
        p.angle_w = self.angle_weight
        p.agent_angle_w = self.agent_angle_weight
        p.proxemics_w = self.proxemics_weight
        p.velocity_feasibility_w = self.velocity_feasibility_weight
        p.obstacle_w = self.obstacle_weight
        p.goal_align_w = self.goal_align_weight
        p.control_horizon = int(self.control_horizon)
        p.parameter_block_length = int(self.parameter_block_length)
        p.max_iterations = int(self.max_iterations)
        p.linear_solver_type = LINEAR_SOLVER[self.linear_solver_type]
        p.fn_tol = self.fn_tol
        p.gradient_tol = self.gradient_tol
        p.param_tol = self.param_tol
        p.desired_linear_vel = self.desired_linear_vel
        p.v_min, p.v_max, p.w_min, p.w_max = self.v_min, self.v_max, self.w_min, self.w_max
        p.fixed_iterations = int(self.fixed_iterations)
        p.tol_needs_successful_step = int(self.tol_needs_successful_step)
        return p

    def dims(self, T: int, has_people: bool = True):
        """(CH, bl, nb, P, M, n_bounded) exactly as src/optimizer.cpp:248-249, 364, 373 derive them."""
        CH = min(self.control_horizon, T)
        bl = min(self.parameter_block_length, CH)
        nb = (CH - 1) // bl + 1
        nfeas = max(0, min(CH // bl, T) - 1)
        M = (8 if has_people else 5) * T + nfeas
        return CH, bl, nb, 2 * nb, M, CH // bl

    def replace(self, **kw) -> "OptimizerParams":
        d = asdict(self)
        d.update(kw)
        return OptimizerParams(**d)

    @staticmethod
    def readme() -> "OptimizerParams":
        """The parameter set of the reference's README.md:66-98 with time_step 0.05 (critics.md / benchmark yaml),
        i.e. the H=18 / bl=6 / T=28 configuration BASELINE.json's metric is quoted on."""
        return OptimizerParams(
            linear_solver_type="DENSE_SCHUR", param_tol=1e-9, fn_tol=1e-5, gradient_tol=1e-8, max_iterations=40,
            control_horizon=18, parameter_block_length=6, current_path_weight=1.0, current_cmds_weight=0.5,
            distance_weight=20.0, social_weight=120.0, velocity_weight=10.0, angle_weight=250.0,
            agent_angle_weight=40.0, velocity_feasibility_weight=5.0, goal_align_weight=10.0, obstacle_weight=0.15,
            proxemics_weight=100.0, time_step=0.05, max_time=1.5)

    @staticmethod
    def params_yaml() -> "OptimizerParams":
        """params/params.yaml:25-56 as shipped (proxemics_weight absent -> code default 90)."""
        return OptimizerParams(
            linear_solver_type="DENSE_SCHUR", param_tol=1e-9, fn_tol=1e-5, gradient_tol=1e-8, max_iterations=40,
            control_horizon=20, parameter_block_length=4, current_path_weight=1.0, current_cmds_weight=0.5,
            distance_weight=50.0, social_weight=700.0, velocity_weight=8.0, angle_weight=180.0,
            agent_angle_weight=0.0, velocity_feasibility_weight=5.0, goal_align_weight=8.0, obstacle_weight=0.2,
            time_step=0.05, max_time=2.0)

    @staticmethod
    def soc_work_obst_benchmark() -> "OptimizerParams":
        """params/soc_work_obst_parameters_in_benchmark.yaml:104-136 as shipped (proxemics_weight absent -> code default
        90; the benchmark's local costmap is 4 m x 4 m at 0.05 m = 80 x 80 cells, :143-149)."""
        return OptimizerParams(
            linear_solver_type="DENSE_SCHUR", param_tol=1e-9, fn_tol=1e-5, gradient_tol=1e-8, max_iterations=40,
            control_horizon=18, parameter_block_length=6, current_path_weight=1.0, current_cmds_weight=0.5,
            distance_weight=20.0, social_weight=120.0, velocity_weight=10.0, angle_weight=250.0,
            agent_angle_weight=40.0, velocity_feasibility_weight=5.0, goal_align_weight=10.0, obstacle_weight=0.13,
            time_step=0.05, max_time=1.5)

    @staticmethod
    def obst_only_benchmark() -> "OptimizerParams":
        """params/obst_only_parameters_in_benchmark.yaml:104-136: the same file with social_weight 0 and
        agent_angle_weight 0 (:129,132) — people are still handed to the optimiser, so the social-work and agent-angle
        rows exist with zero weight and the proxemics row keeps its code default 90."""
        return OptimizerParams.soc_work_obst_benchmark().replace(social_weight=0.0, agent_angle_weight=0.0)

    BENCHMARK_COSTMAP_CELLS = 80  # local costmap of both benchmark files: width = height = 4 m, resolution 0.05 m

    @staticmethod
    def from_yaml(path: str, plugin: str = "FollowPath") -> "OptimizerParams":
        import yaml

        with open(path, "r") as f:
            doc = yaml.load(f, Loader=yaml.SafeLoader)

        def find(node):
            if isinstance(node, dict):
                if plugin in node and isinstance(node[plugin], dict) and "optimizer" in node[plugin]:
                    return node[plugin]
                for v in node.values():
                    r = find(v)
                    if r is not None:
                        return r
            return None

        sec = find(doc)
        if sec is None:
            raise RuntimeError(f"no '{plugin}' section with an 'optimizer' block in {path}")
        opt = dict(sec.get("optimizer", {}))
        weights = dict(opt.pop("weights", {}) or {})
        traj = sec.get("trajectorizer", {}) or {}
        kw = {}
        fields = OptimizerParams.__dataclass_fields__
        for k, v in list(opt.items()) + list(weights.items()):
            if k in fields:
                kw[k] = v
        for k in ("time_step", "max_time"):
            if k in traj:
                kw[k] = traj[k]
        return OptimizerParams(**kw)
