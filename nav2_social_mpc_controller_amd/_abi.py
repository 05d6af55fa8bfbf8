"""ctypes mirror of include/smpc.h (the C ABI of the HIP solver).

Only plain C structs live here; the shared library itself is loaded by `solver.py`.
Struct layouts must stay in lock-step with include/smpc.h (checked by tests/test_abi.py through
`smpc_abi_version` and `sizeof` probes exported by the library).
"""
import ctypes as C

SMPC_ABI_VERSION = 4
SMPC_MAX_BLOCKS = 10

# enum smpc_linear_solver (mirrors OptimizerParams::solver_types, reference optimizer.hpp:71-77)
LINEAR_SOLVER = {
    "DENSE_SCHUR": 0,
    "SPARSE_SCHUR": 1,
    "DENSE_NORMAL_CHOLESKY": 2,
    "DENSE_QR": 3,
    "SPARSE_NORMAL_CHOLESKY": 4,
}

CONVERGENCE, NO_CONVERGENCE, FAILURE = 0, 1, 2
REASONS = ["none", "gradient_tol", "parameter_tol", "function_tol", "min_radius", "max_iterations",
           "invalid_steps", "eval_failed"]

c_double_p = C.POINTER(C.c_double)
c_int32_p = C.POINTER(C.c_int32)
c_uint8_p = C.POINTER(C.c_uint8)


class SmpcParams(C.Structure):
    _fields_ = [
        ("distance_w", C.c_double),
        ("socialwork_w", C.c_double),
        ("velocity_w", C.c_double),
        ("angle_w", C.c_double),
        ("agent_angle_w", C.c_double),
        ("proxemics_w", C.c_double),
        ("velocity_feasibility_w", C.c_double),
        ("obstacle_w", C.c_double),
        ("goal_align_w", C.c_double),
        ("control_horizon", C.c_int),
        ("parameter_block_length", C.c_int),
        ("max_iterations", C.c_int),
        ("linear_solver_type", C.c_int),
        ("fn_tol", C.c_double),
        ("gradient_tol", C.c_double),
        ("param_tol", C.c_double),
        ("desired_linear_vel", C.c_double),
        ("v_min", C.c_double),
        ("v_max", C.c_double),
        ("w_min", C.c_double),
        ("w_max", C.c_double),
        ("fixed_iterations", C.c_int),
        ("tol_needs_successful_step", C.c_int),
    ]


class SmpcSceneBatch(C.Structure):
    _fields_ = [
        ("B", C.c_int32),
        ("T", C.c_int32),
        ("N", C.c_int32),
        ("on_device", C.c_int32),
        ("dt", C.c_double),
        ("pose0", C.c_void_p),
        ("init_params", C.c_void_p),
        ("path_pts", C.c_void_p),
        ("goal_yaw", C.c_void_p),
        ("people", C.c_void_p),
        ("has_people", C.c_void_p),
        ("costmap", C.c_void_p),
        ("costmap_shared", C.c_int32),
        ("size_x", C.c_int32),
        ("size_y", C.c_int32),
        ("costmap_origin", C.c_void_p),
        ("resolution", C.c_double),
        ("people_records", C.c_void_p),
        ("people_aux", C.c_void_p),
        ("order", C.c_void_p),
        ("T_scene", C.c_void_p),
    ]


class SmpcProjectionBatch(C.Structure):
    _fields_ = [
        ("B", C.c_int32),
        ("T", C.c_int32),
        ("N", C.c_int32),
        ("on_device", C.c_int32),
        ("max_time", C.c_float),
        ("time_step", C.c_float),
        ("init_people", C.c_void_p),
        ("robot_path", C.c_void_p),
        ("od_indexes", C.c_void_p),
        ("od_shared", C.c_int32),
        ("od_width", C.c_int32),
        ("od_height", C.c_int32),
        ("od_resolution", C.c_float),
        ("od_origin", C.c_void_p),
    ]


class SmpcPeopleBatch(C.Structure):
    _fields_ = [
        ("B", C.c_int32),
        ("Np", C.c_int32),
        ("N", C.c_int32),
        ("on_device", C.c_int32),
        ("people", C.c_void_p),
        ("count", C.c_void_p),
        ("robot_pose", C.c_void_p),
        ("fov_angle", C.c_double),
        ("costmap_origin", C.c_void_p),
        ("costmap_shared", C.c_int32),
        ("size_x", C.c_int32),
        ("size_y", C.c_int32),
        ("resolution", C.c_double),
    ]


class SmpcMemoryBatch(C.Structure):
    _fields_ = [
        ("prev_path", C.c_void_p),
        ("prev_cmds", C.c_void_p),
        ("valid", C.c_void_p),
        ("length", C.c_void_p),
    ]


class SmpcFormatBatch(C.Structure):
    _fields_ = [
        ("B", C.c_int32),
        ("T", C.c_int32),
        ("path_rows", C.c_int32),
        ("on_device", C.c_int32),
        ("time_step", C.c_float),
        ("current_path_w", C.c_float),
        ("current_cmds_w", C.c_float),
        ("path", C.c_void_p),
        ("cmds", C.c_void_p),
        ("speed", C.c_void_p),
        ("memory", SmpcMemoryBatch),
        ("n_poses", C.c_void_p),
        ("max_poses", C.c_int32),
    ]


class SmpcFormatOut(C.Structure):
    _fields_ = [
        ("robot_status", C.c_void_p),
        ("pose0", C.c_void_p),
        ("init_params", C.c_void_p),
        ("path_pts", C.c_void_p),
        ("goal_yaw", C.c_void_p),
        ("T_scene", C.c_void_p),
    ]


class SmpcTrajectorizeBatch(C.Structure):
    _fields_ = [
        ("B", C.c_int32),
        ("L", C.c_int32),
        ("max_steps", C.c_int32),
        ("on_device", C.c_int32),
        ("omnidirectional", C.c_int32),
        ("desired_linear_vel", C.c_double),
        ("lookahead_dist", C.c_double),
        ("max_angular_vel", C.c_double),
        ("time_step", C.c_double),
        ("plan", C.c_void_p),
        ("plan_len", C.c_void_p),
        ("robot_pose", C.c_void_p),
    ]


class SmpcPlanWindowBatch(C.Structure):
    _fields_ = [
        ("B", C.c_int32),
        ("L", C.c_int32),
        ("on_device", C.c_int32),
        ("reserved", C.c_int32),
        ("max_robot_pose_search_dist", C.c_double),
        ("dist_threshold", C.c_double),
        ("plan", C.c_void_p),
        ("plan_len", C.c_void_p),
        ("plan_start", C.c_void_p),
        ("robot_pose", C.c_void_p),
        ("to_local", C.c_void_p),
    ]


class SmpcTrajectorizeOut(C.Structure):
    _fields_ = [
        ("path", C.c_void_p),
        ("cmds", C.c_void_p),
        ("cmds_vy", C.c_void_p),
        ("n_poses", C.c_void_p),
        ("error", C.c_void_p),
    ]


class SmpcResultBatch(C.Structure):
    _fields_ = [
        ("params", C.c_void_p),
        ("cmds", C.c_void_p),
        ("path", C.c_void_p),
        ("status", C.c_void_p),
        ("reason", C.c_void_p),
        ("iterations", C.c_void_p),
        ("evaluations", C.c_void_p),
        ("initial_cost", C.c_void_p),
        ("final_cost", C.c_void_p),
    ]


class SmpcEvalOut(C.Structure):
    _fields_ = [
        ("residuals", C.c_void_p),
        ("jacobian", C.c_void_p),
        ("cost", C.c_void_p),
        ("gradient", C.c_void_p),
        ("row_order", C.c_int32),
    ]


# Every symbol include/smpc.h declares; tests/test_abi.py checks the built library exports all of them.
EXPORTED_SYMBOLS = [
    "smpc_params_default",
    "smpc_dims",
    "smpc_create",
    "smpc_destroy",
    "smpc_set_stream",
    "smpc_set_solve_share",
    "smpc_solve_slot_width",
    "smpc_solve_batch",
    "smpc_eval_batch",
    "smpc_project_people_batch",
    "smpc_people_to_status_batch",
    "smpc_format_to_optimize_batch",
    "smpc_memory_store_batch",
    "smpc_trajectorize_path_batch",
    "smpc_transform_global_plan_batch",
    "smpc_select_command_batch",
    "smpc_math_probe",
    "smpc_fp64_peak_probe",
    "smpc_stage_people_batch",
    "smpc_last_kernel_ms",
    "smpc_last_error",
    "smpc_abi_version",
]
