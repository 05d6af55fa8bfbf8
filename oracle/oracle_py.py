"""ctypes loader for the CPU oracle (oracle/smpc_oracle.cpp).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
Never import this from nav2_social_mpc_controller_amd/.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from nav2_social_mpc_controller_amd._abi import SmpcEvalOut, SmpcParams, SmpcResultBatch, SmpcSceneBatch

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "libsmpc_oracle.so")
_lib = None


def build():
    subprocess.check_call(["make", "-C", _HERE, "-s"])


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.smpc_oracle_solve_batch.restype = C.c_int
        _lib.smpc_oracle_solve_batch.argtypes = [C.POINTER(SmpcParams), C.POINTER(SmpcSceneBatch),
                                                 C.POINTER(SmpcResultBatch), C.c_int]
        _lib.smpc_oracle_solve_batch2.restype = C.c_int
        _lib.smpc_oracle_solve_batch2.argtypes = [C.POINTER(SmpcParams), C.POINTER(SmpcSceneBatch),
                                                  C.POINTER(SmpcResultBatch), C.c_int, C.c_void_p, C.c_void_p]
        _lib.smpc_oracle_eval_batch.restype = C.c_int
        _lib.smpc_oracle_eval_batch.argtypes = [C.POINTER(SmpcParams), C.POINTER(SmpcSceneBatch), C.c_void_p,
                                                C.POINTER(SmpcEvalOut)]
        _lib.smpc_oracle_solve_trace.restype = C.c_int
        _lib.smpc_oracle_solve_trace.argtypes = [C.POINTER(SmpcParams), C.POINTER(SmpcSceneBatch), C.c_int,
                                                 C.c_void_p, C.c_int, C.c_void_p]
    return _lib


def set_theta_zero_convention(on: bool):
    """False (default): reference-literal sign(theta). True: theta := 0 when both velocities are exactly equal
    (the HIP path's convention; removes the libm-noise dependence of such scenes)."""
    lib().smpc_oracle_set_option(1, 1 if on else 0)


def solve(params, scenes, nthreads=1, theta_zero_convention=False):
    """Solve every scene with the CPU oracle. Returns a dict of numpy arrays (same fields as smpc_result_batch)."""
    set_theta_zero_convention(theta_zero_convention)
    try:
        return _solve(params, scenes, nthreads)
    finally:
        set_theta_zero_convention(False)


def _by_horizon(params, scenes, fn, shapes):
    """Scenes with horizons of their own (SceneBatch.T_scene): every group of equal T_b is handed to the oracle as the
    batch a caller with exactly that horizon would build (arrays cut to T_b + 1 rows / P_b parameters), and the results
    are laid out with the full batch's strides, zeros where a scene has nothing — the layout include/smpc.h documents."""
    B = scenes.B
    out = None
    for Tb in sorted(set(int(t) for t in scenes.T_scene)):
        idx = np.where(scenes.T_scene == Tb)[0]
        P_b = params.dims(Tb, True)[3]
        sub = fn(scenes.cut(idx, Tb, P_b), Tb, idx)
        if out is None:
            out = {k: np.zeros((B,) + shapes[k], v.dtype) if k in shapes else np.zeros((B,) + v.shape[1:], v.dtype)
                   for k, v in sub.items()}
        for k, v in sub.items():
            out[k][(idx,) + tuple(slice(0, n) for n in v.shape[1:])] = v
    return out


def _solve(params, scenes, nthreads=1):
    CH, bl, nb, P, M, _ = params.dims(scenes.T, True)
    B, T = scenes.B, scenes.T
    if getattr(scenes, "T_scene", None) is not None:
        return _by_horizon(params, scenes, lambda sub, Tb, idx: _solve(params, sub, nthreads),
                           {"params": (P,), "cmds": (T + 1, 2), "path": (T + 1, 3)})
    out = {
        "params": np.zeros((B, P)), "cmds": np.zeros((B, T + 1, 2)), "path": np.zeros((B, T + 1, 3)),
        "status": np.zeros(B, np.int32), "reason": np.zeros(B, np.int32), "iterations": np.zeros(B, np.int32),
        "evaluations": np.zeros(B, np.int32), "initial_cost": np.zeros(B), "final_cost": np.zeros(B),
    }
    rb = SmpcResultBatch()
    for k, v in out.items():
        setattr(rb, k, v.ctypes.data)
    cp, sb = params.to_c(), scenes.to_c()
    events = np.zeros(B, np.int32)
    marginal = np.zeros(B, np.int32)
    rc = lib().smpc_oracle_solve_batch2(C.byref(cp), C.byref(sb), C.byref(rb), int(nthreads), events.ctypes.data,
                                        marginal.ctypes.data)
    if rc != 0:
        raise RuntimeError(f"smpc_oracle_solve_batch failed: {rc}")
    # diagnostic: evaluations whose sign(theta) was decided by libm rounding noise (robot stopped beside a standing
    # person); the reference's own result is not reproducible across libm builds for such scenes
    out["sign_noise_events"] = events
    # diagnostic: accept / terminate / Armijo decisions taken with a margin below 1e-12 of the cost (rounding noise)
    out["marginal_decisions"] = marginal
    return out


def evaluate(params, scenes, x, jacobian=True):
    """Residuals / Jacobian / cost / gradient of every scene at parameters x [B,P]."""
    CH, bl, nb, P, M, _ = params.dims(scenes.T, True)
    B = scenes.B
    x = np.ascontiguousarray(x, dtype=np.float64)
    assert x.shape == (B, P)
    if getattr(scenes, "T_scene", None) is not None:  # reference row order, rows / columns a scene does not have are zero
        return _by_horizon(params, scenes,
                           lambda sub, Tb, idx: evaluate(params, sub, x[idx][:, :params.dims(Tb, True)[3]], jacobian),
                           {"residuals": (M,), "jacobian": (M, P), "gradient": (P,)})
    out = {"residuals": np.zeros((B, M)), "cost": np.zeros(B)}
    if jacobian:
        out["jacobian"] = np.zeros((B, M, P))
        out["gradient"] = np.zeros((B, P))
    eo = SmpcEvalOut()
    for k, v in out.items():
        setattr(eo, k, v.ctypes.data)
    cp, sb = params.to_c(), scenes.to_c()
    rc = lib().smpc_oracle_eval_batch(C.byref(cp), C.byref(sb), x.ctypes.data, C.byref(eo))
    if rc != 0:
        raise RuntimeError(f"smpc_oracle_eval_batch failed: {rc}")
    return out


TRACE_COLS = ["iter", "cost", "cost_change", "gradient_max_norm", "step_norm", "rho", "radius", "ls_evals", "accepted"]


def trace(params, scenes, scene=0, max_rows=256):
    rows = np.zeros((max_rows, 9))
    cp, sb = params.to_c(), scenes.to_c()
    n = lib().smpc_oracle_solve_trace(C.byref(cp), C.byref(sb), int(scene), rows.ctypes.data, max_rows, None)
    if n < 0:
        raise RuntimeError(f"smpc_oracle_solve_trace failed: {n}")
    return rows[:min(n, max_rows)]


OP_NAMES = ["add", "mul", "div", "sqrt", "exp", "sincos", "atan2"]
_count_lib = None


def count_ops(params, scenes, scene=0, x=None):
    """Scalar FP64 operations of the Jet passes of ONE Jacobian evaluation in the reference's formulation
    (DynamicAutoDiffCostFunction on ceres::Jet<double, 4>), counted by an instrumented build of the oracle
    (oracle/_build/libsmpc_oracle_count.so). Returns a dict name -> count."""
    global _count_lib
    if _count_lib is None:
        path = os.path.join(_HERE, "_build", "libsmpc_oracle_count.so")
        if not os.path.exists(path):
            build()
        _count_lib = C.CDLL(path)
        _count_lib.smpc_oracle_count_ops.restype = C.c_int
        _count_lib.smpc_oracle_count_ops.argtypes = [C.POINTER(SmpcParams), C.POINTER(SmpcSceneBatch), C.c_int, C.c_void_p, C.c_void_p]
    x = np.ascontiguousarray(scenes.init_params[scene] if x is None else x, dtype=np.float64)
    out = np.zeros(7, dtype=np.int64)
    cp, sb = params.to_c(), scenes.to_c()
    rc = _count_lib.smpc_oracle_count_ops(C.byref(cp), C.byref(sb), int(scene), x.ctypes.data, out.ctypes.data)
    if rc != 0:
        raise RuntimeError(f"smpc_oracle_count_ops failed: {rc}")
    return dict(zip(OP_NAMES, (int(v) for v in out)))
