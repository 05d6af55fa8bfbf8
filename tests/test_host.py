"""Host-side logic: parameter mirror, YAML reader, scene generator, shard maths (CPU only)."""
import os

import numpy as np
import pytest

from nav2_social_mpc_controller_amd.params import OptimizerParams
from nav2_social_mpc_controller_amd.scenes import SceneBatch, make_scenes, uniform
from nav2_social_mpc_controller_amd import dist as D

YAML = """
controller_server:
  ros__parameters:
    controller_frequency: 20.0
    FollowPath:
      plugin: "nav2_social_mpc_controller::SocialMPCController"
      trajectorizer:
        omnidirectional: false
        time_step: 0.05
        max_time: 1.5
      optimizer:
        linear_solver_type: "DENSE_SCHUR"
        param_tol: 1.0e-9
        fn_tol: 1.0e-5
        gradient_tol: 1.0e-8
        max_iterations: 40
        control_horizon: 18
        parameter_block_length: 6
        discretization: 1
        debug_optimizer: false
        current_path_weight: 1.0
        current_cmds_weight: 0.5
        weights:
          distance_weight: 20.0
          social_weight: 120.0
          velocity_weight: 10.0
          angle_weight: 250.0
          agent_angle_weight: 40.0
          velocity_feasibility_weight: 5.0
          goal_align_weight: 10.0
          obstacle_weight: 0.13
"""


def test_yaml_reader_handles_a_nav2_shaped_file(tmp_path):
    f = tmp_path / "p.yaml"
    f.write_text(YAML)
    p = OptimizerParams.from_yaml(str(f))
    assert p.linear_solver_type == "DENSE_SCHUR" and p.control_horizon == 18 and p.parameter_block_length == 6
    assert p.obstacle_weight == 0.13 and p.social_weight == 120.0
    assert p.proxemics_weight == 90.0          # absent in the file -> reference code default (src/optimizer.cpp:67)
    assert p.max_time == 1.5 and p.rollout_steps == 28


def test_benchmark_presets_are_the_shipped_parameter_files():
    """OptimizerParams.soc_work_obst_benchmark / obst_only_benchmark against what the YAML reader finds in the reference's
    own params/*_in_benchmark.yaml:104-136 (read here, in the build container; the GPU box does not have the reference)."""
    ref = "/root/reference/params"
    a, b = OptimizerParams.soc_work_obst_benchmark(), OptimizerParams.obst_only_benchmark()
    assert (a.social_weight, a.agent_angle_weight, a.obstacle_weight, a.proxemics_weight) == (120.0, 40.0, 0.13, 90.0)
    assert (b.social_weight, b.agent_angle_weight, b.obstacle_weight, b.proxemics_weight) == (0.0, 0.0, 0.13, 90.0)
    assert a.dims(a.rollout_steps) == (18, 6, 3, 6, 226, 3) and a.rollout_steps == 28
    if not os.path.isdir(ref):
        pytest.skip("reference tree not present on this machine")
    assert OptimizerParams.from_yaml(os.path.join(ref, "soc_work_obst_parameters_in_benchmark.yaml")) == a
    assert OptimizerParams.from_yaml(os.path.join(ref, "obst_only_parameters_in_benchmark.yaml")) == b
    assert OptimizerParams.from_yaml(os.path.join(ref, "params.yaml")) == OptimizerParams.params_yaml()


def test_time_step_is_a_widened_float():
    p = OptimizerParams.readme()
    assert p.dt == float(np.float32(0.05)) and p.dt != 0.05


def test_rollout_steps_follow_format_to_optimize():
    assert OptimizerParams.readme().rollout_steps == 28            # round(1.5/0.05)=30 -> 29 poses -> 28 velocities
    assert OptimizerParams.params_yaml().rollout_steps == 38
    assert OptimizerParams.readme().replace(time_step=0.1).rollout_steps == 13


def test_rng_is_counter_based():
    ids = np.arange(10, 20)
    a = uniform(1, ids, 3, 4)
    b = uniform(1, ids[5:], 3, 4)
    assert np.array_equal(a[5:], b)
    assert a.min() >= 0.0 and a.max() < 1.0
    assert not np.array_equal(uniform(2, ids, 3, 4), a)


def test_scene_shards_regenerate_identically():
    p = OptimizerParams.readme()
    whole = make_scenes(p, 12, 4, map_cells=60, seed=77)
    for r in range(3):
        lo, hi = D.shard_range(12, r, 3)
        part = make_scenes(p, hi - lo, 4, map_cells=60, seed=77, first_scene=lo)
        for k in ("pose0", "init_params", "path_pts", "goal_yaw", "people", "costmap", "costmap_origin"):
            assert np.array_equal(getattr(part, k), getattr(whole, k)[lo:hi]), k


def test_scene_shapes_and_quirks():
    p = OptimizerParams.readme()
    sc = make_scenes(p, 5, 3, n_valid=1, map_cells=60)
    assert sc.people.shape == (5, 29, 6, 3)
    assert np.all(sc.people[:, :, 3, 1:] == -1.0) and np.all(sc.people[:, :, [0, 1, 2, 4, 5], 1:] == 0.0)
    # aliasing quirk (src/optimizer.cpp:254-261): block 0 starts from the current twist, later blocks from cmds
    assert np.all(sc.init_params[:, 2] == 0.6) and np.all(sc.init_params[:, 4] == 0.6)
    assert np.all(sc.init_params[:, 0] <= 0.6)
    assert sc.costmap.dtype == np.uint8 and sc.costmap.max() == 254


def test_scene_save_load_roundtrip(tmp_path):
    p = OptimizerParams.readme()
    sc = make_scenes(p, 3, 4, map_cells=40)
    f = str(tmp_path / "s.npz")
    sc.save(f)
    sc2 = SceneBatch.load(f)
    for k in ("pose0", "init_params", "path_pts", "goal_yaw", "people", "has_people", "costmap", "costmap_origin"):
        assert np.array_equal(getattr(sc, k), getattr(sc2, k))
    assert (sc2.T, sc2.N, sc2.dt, sc2.resolution) == (sc.T, sc.N, sc.dt, sc.resolution)


def test_shard_ranges_partition_the_batch():
    for total, world in ((65536, 8), (10, 3), (7, 8)):
        covered = []
        for r in range(world):
            lo, hi = D.shard_range(total, r, world)
            covered += list(range(lo, hi))
        assert covered == list(range(total))
    assert D.weak_shard(8192, 3) == (3 * 8192, 4 * 8192)
