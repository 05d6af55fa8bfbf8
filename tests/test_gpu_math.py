"""csrc/smpc_math.hpp on the device (through smpc_math_probe): the table-driven exp / atan2 / sincos and the refined
reciprocal / rsqrt of the sweep against libm, plus the accuracy of the hardware estimates the refinements start from
(their sizing assumes >= 22 good bits; measured here so that a different part would fail loudly)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def solver():
    from nav2_social_mpc_controller_amd.params import OptimizerParams
    from nav2_social_mpc_controller_amd.solver import BatchSolver
    return BatchSolver(OptimizerParams.readme(), device=0)


def ulps(got, want):
    want = np.asarray(want, dtype=np.float64)
    return np.abs(got - want) / np.spacing(np.abs(want))


def test_exp(solver):
    rng = np.random.default_rng(1)
    x = np.concatenate([-rng.uniform(0, 60, 400000), -rng.uniform(0, 745, 100000), -10.0 ** rng.uniform(-20, 0, 20000),
                        np.array([0.0, -0.0, -1e-300, -745.0, -746.0, -800.0, -1e300, -np.inf])])
    o, _ = solver.math_probe(0, x)
    want = np.exp(x)
    normal = want > 1e-300
    assert ulps(o[normal], want[normal]).max() <= 1.0
    assert np.all(o[~normal] <= 1e-300) and np.all(o[~normal] >= 0.0)


def test_atan2_of_directions(solver):
    rng = np.random.default_rng(2)
    ang = np.concatenate([rng.uniform(-np.pi, np.pi, 600000), np.array([0.0, np.pi / 4, np.pi / 2, 3 * np.pi / 4, np.pi, -np.pi / 2]),
                          rng.choice([-1, 1], 20000) * 10.0 ** rng.uniform(-12, -1, 20000),
                          np.pi - 10.0 ** rng.uniform(-12, -1, 20000)])
    scale = 1.0 + 1e-15 * rng.standard_normal(len(ang))
    y, x = np.sin(ang) * scale, np.cos(ang) * scale
    o, _ = solver.math_probe(1, y, x)
    u = ulps(o, np.arctan2(y, x))
    assert u.max() <= 2.0 and (u > 1.0).mean() < 1e-3


def test_atan2_of_unit_vectors(solver):
    """atan2_unit on the device (node table in LDS): the same absolute bound as on the host (tests/test_math.py)."""
    rng = np.random.default_rng(12)
    ang = np.concatenate([rng.uniform(-np.pi, np.pi, 600000),
                          np.array([0.0, np.pi / 4, np.pi / 2, 3 * np.pi / 4, np.pi, -np.pi / 2, -np.pi / 4]),
                          rng.choice([-1, 1], 20000) * 10.0 ** rng.uniform(-12, -1, 20000),
                          np.pi - 10.0 ** rng.uniform(-12, -1, 20000),
                          np.arcsin((np.arange(48) + 0.5) / 64.0)])
    scale = 1.0 + 5e-16 * rng.standard_normal(len(ang))
    y, x = np.sin(ang) * scale, np.cos(ang) * scale
    o, _ = solver.math_probe(8, y, x)
    want = np.arctan2(y, x)
    assert np.abs(o - want).max() <= 6e-16
    small = np.abs(want) < 0.0156
    assert (np.abs(o[small] - want[small]) <= 4e-15 * np.abs(want[small])).all()


def test_sincos(solver):
    rng = np.random.default_rng(3)
    x = np.concatenate([rng.uniform(-20, 20, 400000), rng.uniform(-1e5, 1e5, 100000), np.arange(-40, 41) * (np.pi / 4)])
    s, c = solver.math_probe(2, x)
    assert np.abs(s - np.sin(x)).max() <= 2.3e-16
    assert np.abs(c - np.cos(x)).max() <= 2.3e-16


def test_rsqrt_division_and_the_hardware_estimates(solver):
    rng = np.random.default_rng(4)
    x = 10.0 ** rng.uniform(-12, 6, 400000)
    o, _ = solver.math_probe(3, x)
    assert ulps(o, 1.0 / np.sqrt(x)).max() <= 2.0
    a = rng.uniform(0, 1, 400000)
    b = np.maximum(a, rng.uniform(0.5, 1, 400000))
    q, _ = solver.math_probe(4, a, b)
    assert ulps(q[a > 0], (a / b)[a > 0]).max() <= 1.0
    rcp, _ = solver.math_probe(5, x)
    rsq, _ = solver.math_probe(6, x)
    e_rcp = np.abs(rcp * x - 1.0).max()
    e_rsq = np.abs(rsq * np.sqrt(x) - 1.0).max()
    print(f"hardware estimates: v_rcp_f64 max rel err {e_rcp:.3e}, v_rsq_f64 max rel err {e_rsq:.3e}")
    assert e_rcp <= 2.0 ** -22 and e_rsq <= 2.0 ** -22
