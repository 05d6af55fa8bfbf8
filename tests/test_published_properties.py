"""Anchors to PUBLISHED properties of the dependency the reference delegates to (Ceres Solver, absent here — DESIGN.md §2:
parity with the reference binary is unpinned). Ceres' own unit tests state two properties that any faithful restatement
must have, and that a wrong kernel convention would break:

* `internal/ceres/cubic_interpolation_test.cc` (BiCubicInterpolator: ZeroFunction .. Degree22Function): data sampled
  from a polynomial of degree <= 2 in each variable is reproduced EXACTLY by the interpolant, value and both partial
  derivatives — true of the Catmull-Rom kernel (a = -1/2) and of no other cubic convolution kernel; `Grid2D` clamps
  indices outside the grid to its edge.
* `internal/ceres/polynomial_test.cc` (FindInterpolatingPolynomial / MinimizePolynomial): the polynomial through samples
  (values, optionally gradients) of a polynomial of matching degree IS that polynomial, and its minimiser on an interval
  is the interior critical point with the smallest value or an interval end.

Checked here on the Python restatement directly, on the C++ oracle through the obstacle critic's rows (a5:
critics/obstacle_cost_function.hpp:154-161) and — `-m gpu` — on the HIP path through the same rows."""
import numpy as np
import pytest
import torch

from nav2_social_mpc_controller_amd.params import OptimizerParams
from nav2_social_mpc_controller_amd.scenes import make_scenes
from oracle import pyref

# degree <= 2 in each variable with integer values in 0..255 on a 16 x 16 grid (u8 costmaps): p(col, row)
POLYS = {
    "constant": lambda x, y: 7 + 0 * x + 0 * y,
    "linear_x": lambda x, y: 3 + 11 * x + 0 * y,
    "linear_y": lambda x, y: 5 + 0 * x + 13 * y,
    "bilinear": lambda x, y: x * y + y,
    "quadratic_x": lambda x, y: 9 + (x * x + x) / 2 + 2 * y,
    "quadratic_y": lambda x, y: 1 + x + y * y,
    "quadratic_both": lambda x, y: (x * x + x) / 2 + (y * y + y) / 2,
}
SIZE = 16


def _grid(name):
    y, x = np.mgrid[0:SIZE, 0:SIZE]
    g = np.asarray(POLYS[name](x, y), dtype=np.float64)
    assert g.min() >= 0 and g.max() <= 255 and np.all(g == np.round(g))
    return g.astype(np.uint8)


def _poly_and_grad(name, c, r):
    """p, dp/dcol, dp/drow at real (col, row) by exact differentiation of the integer polynomial's real extension."""
    ct, rt = torch.tensor(float(c), dtype=torch.float64, requires_grad=True), torch.tensor(float(r), dtype=torch.float64, requires_grad=True)
    v = POLYS[name](ct, rt) + 0.0 * (ct + rt)
    gc, gr = torch.autograd.grad(v, (ct, rt))
    return float(v.detach()), float(gc), float(gr)


@pytest.mark.parametrize("name", sorted(POLYS))
def test_python_restatement_bicubic_reproduces_polynomial_data(name):
    g = _grid(name)
    rng = np.random.default_rng(3)
    for _ in range(40):
        r, c = rng.uniform(1.0, SIZE - 2.001, 2)   # the whole 4 x 4 patch inside the grid
        rt, ct = torch.tensor(r, dtype=torch.float64, requires_grad=True), torch.tensor(c, dtype=torch.float64, requires_grad=True)
        f = pyref._bicubic(g, rt, ct)
        dfr, dfc = torch.autograd.grad(f, (rt, ct), allow_unused=True)
        want, wc, wr = _poly_and_grad(name, c, r)
        assert abs(float(f.detach()) - want) <= 1e-11 * max(1.0, abs(want))
        assert abs((0.0 if dfc is None else float(dfc)) - wc) <= 1e-10 * max(1.0, abs(wc))
        assert abs((0.0 if dfr is None else float(dfr)) - wr) <= 1e-10 * max(1.0, abs(wr))


def test_python_restatement_grid_clamps_to_its_edge():
    """Grid2D::GetValue clamps row / column indices: outside the grid the interpolant is that of the edge-replicated data."""
    rng = np.random.default_rng(4)
    g = rng.integers(0, 256, (SIZE, SIZE)).astype(np.uint8)
    pad = 6
    big = np.pad(g, pad, mode="edge")
    for _ in range(60):
        r, c = rng.uniform(-3.5, SIZE + 2.5, 2)
        a = pyref._bicubic(g, torch.tensor(r, dtype=torch.float64), torch.tensor(c, dtype=torch.float64))
        b = pyref._bicubic(big, torch.tensor(r + pad, dtype=torch.float64), torch.tensor(c + pad, dtype=torch.float64))
        assert abs(float(a) - float(b)) <= 1e-12 * max(1.0, abs(float(b)))   # (the shifted argument's fraction rounds differently)


def _obstacle_only_case(name):
    """Four robots driving straight (angular velocities 0) over a costmap holding polynomial `name`; only the obstacle
    critic weighted. Returns (params, scenes, x, expected residual [B, T], expected d residual / d v_block [B, T, nb])."""
    prm = OptimizerParams.readme().replace(distance_weight=0.0, angle_weight=0.0, velocity_weight=0.0, social_weight=0.0,
                                           agent_angle_weight=0.0, proxemics_weight=0.0, goal_align_weight=0.0,
                                           velocity_feasibility_weight=0.0, obstacle_weight=1.0)
    res = 0.25
    sc = make_scenes(prm, 4, 1, people_present=False, map_cells=SIZE, resolution=res)
    sc.costmap = np.broadcast_to(_grid(name), (sc.B, SIZE, SIZE)).copy()
    T = sc.T
    CH, bl, nb, P, M, _ = prm.dims(T, True)
    yaw = np.array([0.3, -1.1, 2.0, 0.0])
    v = np.array([0.45, 0.3, 0.5, 0.2])
    sc.pose0 = np.stack([np.full(4, 0.0), np.full(4, 0.0), yaw], axis=1)
    # origin such that every front point keeps its 4 x 4 patch inside the grid: start near cell (6, 6) + heading margin
    sc.costmap_origin = np.stack([-(6.0 + 2.5 * (np.cos(yaw) < 0)) * res, -(6.0 + 2.5 * (np.sin(yaw) < 0)) * res], axis=1)
    x = np.zeros((sc.B, P))
    x[:, 0::2] = v[:, None]
    steps = np.arange(1, T + 1)
    px = sc.pose0[:, 0:1] + v[:, None] * sc.dt * steps[None, :] * np.cos(yaw)[:, None] + 0.25 * np.cos(yaw)[:, None]
    py = sc.pose0[:, 1:2] + v[:, None] * sc.dt * steps[None, :] * np.sin(yaw)[:, None] + 0.25 * np.sin(yaw)[:, None]
    col = (px - sc.costmap_origin[:, 0:1]) / res
    row = (py - sc.costmap_origin[:, 1:2]) / res
    assert col.min() >= 1.0 and col.max() <= SIZE - 2.001 and row.min() >= 1.0 and row.max() <= SIZE - 2.001
    want = np.zeros((sc.B, T))
    dv = np.zeros((sc.B, T, nb))
    for b in range(sc.B):
        for i in range(T):
            p, gc, gr = _poly_and_grad(name, col[b, i], row[b, i])
            want[b, i] = p
            # pose_{i+1} = pose0 + sum_{j <= i} v_{block(j)} dt (cos, sin)(yaw): the steps block q drives up to i
            for q in range(nb):
                lo = q * bl
                hi = (q + 1) * bl if q < nb - 1 else T
                cnt = max(0, min(i + 1, hi) - lo)
                dv[b, i, q] = (gc * np.cos(yaw[b]) + gr * np.sin(yaw[b])) * sc.dt * cnt / res
    nfeas = max(min(CH // bl, T) - 1, 0)
    rows = np.array([5 * i + min(max(i - 1, 0), nfeas) + 4 for i in range(T)])   # the obstacle row of step i (no people)
    return prm, sc, x, want, dv, rows


def _check_rows(ev, case, rtol):
    prm, sc, x, want, dv, rows = case
    r = ev["residuals"][:, rows]
    assert np.abs(r - want).max() <= rtol * max(1.0, np.abs(want).max())
    J = ev["jacobian"][:, rows, :]
    assert np.abs(J[:, :, 0::2] - dv).max() <= 10 * rtol * max(1.0, np.abs(dv).max())
    # nothing else is weighted: every other row is zero
    other = np.ones(ev["residuals"].shape[1], bool)
    other[rows] = False
    assert np.abs(ev["residuals"][:, other]).max() == 0.0


@pytest.mark.parametrize("name", sorted(POLYS))
def test_oracle_obstacle_rows_reproduce_polynomial_costmaps(name):
    from oracle import oracle_py as O
    case = _obstacle_only_case(name)
    _check_rows(O.evaluate(case[0], case[1], case[2]), case, 1e-11)


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(POLYS))
def test_device_obstacle_rows_reproduce_polynomial_costmaps(name):
    from nav2_social_mpc_controller_amd.solver import BatchSolver
    case = _obstacle_only_case(name)
    _check_rows(BatchSolver(case[0]).evaluate(case[1], case[2]), case, 1e-11)


def test_interpolating_polynomial_and_its_minimiser():
    """polynomial_test.cc: the interpolant through samples of a polynomial of matching degree is that polynomial; its
    minimiser on [lo, hi] is the best of the interior critical points and the two ends (the line search's step, A.8)."""
    rng = np.random.default_rng(11)

    def true_min(poly, lo, hi):
        cand = [lo, hi] + [float(np.real(z)) for z in np.roots(np.polyder(poly)) if abs(np.imag(z)) < 1e-12 and lo <= np.real(z) <= hi]
        vals = [np.polyval(poly, c) for c in cand]
        return cand[int(np.argmin(vals))], min(vals)

    for deg, pts in ((2, [(0.0, True), (1.0, False)]), (3, [(0.0, True), (0.8, True)]), (5, [(0.0, True), (0.5, True), (1.0, True)])):
        for _ in range(50):
            poly = rng.standard_normal(deg + 1)
            samples = [(xv, float(np.polyval(poly, xv)), float(np.polyval(np.polyder(poly), xv)) if g else None) for xv, g in pts]
            lo, hi = sorted(rng.uniform(0.02, 0.9, 2))
            got = pyref._interp_min(samples, lo, hi)
            want_x, want_v = true_min(poly, lo, hi)
            assert lo <= got <= hi
            assert np.polyval(poly, got) <= want_v + 1e-9 * max(1.0, abs(want_v))
