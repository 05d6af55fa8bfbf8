"""The C-ABI library loads, exports every symbol include/smpc.h declares, and its struct layouts match the
ctypes mirror. No compute calls (no GPU needed)."""
import ctypes as C
import os
import re
import subprocess
import sys

import pytest

from nav2_social_mpc_controller_amd import _abi
from nav2_social_mpc_controller_amd import solver as S

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "smpc.h")


def _declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = re.findall(r"\b(smpc_[a-z0-9_]+)\s*\(", src)
    return sorted(set(names))


def test_header_and_python_symbol_lists_agree():
    assert _declared_functions() == sorted(_abi.EXPORTED_SYMBOLS)


def test_library_exports_every_declared_symbol():
    assert os.path.exists(S.LIB_PATH), "run __graft_entry__.build() first"
    out = subprocess.check_output(["nm", "-D", "--defined-only", S.LIB_PATH], text=True)
    exported = {line.split()[-1] for line in out.splitlines() if line.strip()}
    for name in _declared_functions():
        assert name in exported, f"{name} declared in include/smpc.h but not exported by libsmpc_hip.so"


def test_library_loads_and_reports_abi_version():
    lib = S.load_library()
    assert lib.smpc_abi_version() == _abi.SMPC_ABI_VERSION


def test_struct_layouts_match_the_c_header(tmp_path):
    prog = tmp_path / "layout.c"
    prog.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "smpc.h"\nint main(void){\n'
                    'printf("%zu %zu %zu %zu\\n", sizeof(smpc_params), sizeof(smpc_scene_batch), sizeof(smpc_result_batch), sizeof(smpc_eval_batch_out));\n'
                    'printf("%zu %zu %zu %zu\\n", offsetof(smpc_params, fn_tol), offsetof(smpc_params, fixed_iterations), offsetof(smpc_scene_batch, costmap_origin), offsetof(smpc_scene_batch, resolution));\n'
                    'return 0;}\n')
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(prog), "-o", str(exe)])
    l1, l2 = subprocess.check_output([str(exe)], text=True).splitlines()
    sizes = [int(v) for v in l1.split()]
    offs = [int(v) for v in l2.split()]
    assert sizes == [C.sizeof(_abi.SmpcParams), C.sizeof(_abi.SmpcSceneBatch), C.sizeof(_abi.SmpcResultBatch), C.sizeof(_abi.SmpcEvalOut)]
    assert offs == [_abi.SmpcParams.fn_tol.offset, _abi.SmpcParams.fixed_iterations.offset,
                    _abi.SmpcSceneBatch.costmap_origin.offset, _abi.SmpcSceneBatch.resolution.offset]


def test_next_row_struct_layouts_match_the_c_header(tmp_path):
    """ctypes mirrors of the SURVEY §8(f) structs: sizes and a late field's offset each."""
    probes = [("smpc_projection_batch", _abi.SmpcProjectionBatch, "od_origin"),
              ("smpc_people_batch", _abi.SmpcPeopleBatch, "resolution"),
              ("smpc_memory_batch", _abi.SmpcMemoryBatch, "valid"),
              ("smpc_format_batch", _abi.SmpcFormatBatch, "memory"),
              ("smpc_format_out", _abi.SmpcFormatOut, "goal_yaw"),
              ("smpc_trajectorize_batch", _abi.SmpcTrajectorizeBatch, "robot_pose"),
              ("smpc_trajectorize_out", _abi.SmpcTrajectorizeOut, "error"),
              ("smpc_plan_window_batch", _abi.SmpcPlanWindowBatch, "to_local")]
    body = "".join(f'printf("%zu %zu\\n", sizeof({c}), offsetof({c}, {f}));\n' for c, _, f in probes)
    prog = tmp_path / "layout2.c"
    prog.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "smpc.h"\nint main(void){\n' + body + 'return 0;}\n')
    exe = tmp_path / "layout2"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(prog), "-o", str(exe)])
    lines = subprocess.check_output([str(exe)], text=True).splitlines()
    for (cname, py, field), line in zip(probes, lines):
        size, off = (int(v) for v in line.split())
        assert size == C.sizeof(py), cname
        assert off == getattr(py, field).offset, (cname, field)


def test_params_default_matches_reference_code_defaults():
    lib = S.load_library()
    p = _abi.SmpcParams()
    lib.smpc_params_default(C.byref(p))
    from nav2_social_mpc_controller_amd.params import OptimizerParams
    q = OptimizerParams().to_c()
    for name, _ in _abi.SmpcParams._fields_:
        assert getattr(p, name) == getattr(q, name), name


def test_dims_follow_the_reference_rules():
    lib = S.load_library()
    from nav2_social_mpc_controller_amd.params import OptimizerParams
    cases = [(OptimizerParams.readme(), 28, (18, 6, 3, 6, 226, 3)),          # H18/bl6, T=28: P=6, M=226
             (OptimizerParams.params_yaml(), 38, (20, 4, 5, 10, 308, 5)),    # params.yaml: P=10, M=308
             (OptimizerParams.params_yaml().replace(control_horizon=18), 38, (18, 4, 5, 10, 307, 4)),
             (OptimizerParams.readme().replace(time_step=0.1), 13, (13, 6, 3, 6, 105, 2))]  # bl does not divide CH
    for prm, T, want in cases:
        vals = [C.c_int() for _ in range(6)]
        cp = prm.to_c()
        rc = lib.smpc_dims(C.byref(cp), T, 1, *[C.byref(v) for v in vals])
        assert rc == 0
        assert tuple(v.value for v in vals) == want
        assert prm.dims(T, True) == want


def test_invalid_solver_type_is_rejected_like_the_reference():
    from nav2_social_mpc_controller_amd.params import OptimizerParams
    with pytest.raises(RuntimeError, match="linear_solver_type"):
        OptimizerParams(linear_solver_type="NOT_A_SOLVER")


@pytest.mark.skipif(os.path.exists("/dev/kfd"), reason="checks the no-GPU failure mode")
def test_create_fails_loudly_without_a_gpu():
    """No CPU fallback behind the ABI: creating a solver on a box without a HIP device must raise."""
    from nav2_social_mpc_controller_amd.params import OptimizerParams
    with pytest.raises(S.SmpcError, match="no HIP device|hip"):
        S.BatchSolver(OptimizerParams.readme())


def test_block_limit_matches_the_header():
    """smpc_dims reports nb for any shape; the header's SMPC_MAX_BLOCKS is what the library instantiates (nb 1..10)."""
    lib = S.load_library()
    src = open(HEADER).read()
    assert int(re.search(r"#define SMPC_MAX_BLOCKS (\d+)", src).group(1)) == 10
    p = _abi.SmpcParams()
    lib.smpc_params_default(C.byref(p))
    nb = C.c_int()
    p.control_horizon, p.parameter_block_length = 33, 3
    assert lib.smpc_dims(C.byref(p), 40, 1, None, None, C.byref(nb), None, None, None) == 0 and nb.value == 11
    hip_src = open(os.path.join(ROOT, "nav2_social_mpc_controller_amd", "csrc", "smpc_hip.hip")).read()
    assert "case 10: return pick_w<10>" in hip_src and "case 11" not in hip_src


def test_absurd_iteration_caps_are_refused_before_anything_is_launched():
    """A persistent wave runs until its scenes stop: smpc_create refuses max_iterations outside 0 .. 100000 (the check
    comes before the device query, so it is visible on a box without a GPU too)."""
    from nav2_social_mpc_controller_amd.params import OptimizerParams
    src = open(HEADER).read()
    cap = int(re.search(r"#define SMPC_MAX_LM_ITERATIONS (\d+)", src).group(1))
    assert cap == 100000
    for bad in (-1, cap + 1, 2 ** 31 - 1):
        with pytest.raises(S.SmpcError, match="max_iterations"):
            S.BatchSolver(OptimizerParams.readme().replace(max_iterations=bad))
