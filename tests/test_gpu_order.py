"""smpc_scene_batch.order: the queue order of the persistent solve kernel is a scheduling hint — every scene is solved
by its own lanes from its own inputs, so the results must not depend on it, bit for bit."""
import numpy as np
import pytest

from nav2_social_mpc_controller_amd.params import OptimizerParams
from nav2_social_mpc_controller_amd.scenes import make_scenes


@pytest.mark.gpu
def test_results_do_not_depend_on_the_queue_order():
    from nav2_social_mpc_controller_amd.solver import BatchSolver

    prm = OptimizerParams.readme()
    B = 777
    sc = make_scenes(prm, B, 8)
    s = BatchSolver(prm)
    base = s.solve(sc)
    rng = np.random.default_rng(5)
    for order in (np.argsort(-base["evaluations"], kind="stable"), rng.permutation(B), np.arange(B)[::-1]):
        got = s.solve(sc, order=order)
        for k in base:
            assert np.array_equal(base[k], got[k]), k


@pytest.mark.gpu
def test_a_host_order_that_is_not_a_permutation_is_refused():
    from nav2_social_mpc_controller_amd.solver import BatchSolver, SmpcError

    prm = OptimizerParams.readme()
    sc = make_scenes(prm, 16, 3)
    s = BatchSolver(prm)
    for bad in (np.zeros(16, np.int32), np.arange(16)[::-1] + 1, np.r_[np.arange(15), -1]):
        with pytest.raises(SmpcError, match="permutation"):
            s.solve(sc, order=bad)


@pytest.mark.gpu
def test_device_order_entries_outside_the_batch_are_skipped():
    """Device arrays cannot be checked on the host, but a wrong entry must not become an out-of-bounds scene index: it
    is skipped, and the scene such an order leaves out must not look solved: its status reads SMPC_NOT_SOLVED (-1), not
    whatever an earlier call left there. A repeated entry solves that scene twice (same result) and leaves another out."""
    import torch

    from nav2_social_mpc_controller_amd.solver import BatchSolver

    prm = OptimizerParams.readme()
    B = 64
    sc = make_scenes(prm, B, 3)
    s = BatchSolver(prm)
    base = s.solve(sc)
    sb, tens = sc.to_device()
    rb, rt = s.alloc_results(B, sc.T)
    rt["status"].fill_(-7)
    order = torch.arange(B, dtype=torch.int32, device="cuda:0").flip(0).contiguous()
    order[3] = B + 1000     # scene B-4 is never handed out
    order[10] = -5          # scene B-11 neither
    order[20] = order[21]   # scene B-21 is replaced by a second B-22
    sb.order = order.data_ptr()
    s.solve_device(sb, rb)
    torch.cuda.synchronize()
    st = rt["status"].cpu().numpy()
    skipped = np.zeros(B, bool)
    skipped[[B - 4, B - 11, B - 21]] = True
    assert (st[skipped] == -1).all()
    assert np.array_equal(st[~skipped], base["status"][~skipped])
    assert np.array_equal(rt["cmds"].cpu().numpy()[~skipped], base["cmds"][~skipped])


@pytest.mark.gpu
def test_solve_share_changes_the_grid_not_the_results():
    """smpc_set_solve_share sizes the persistent grid for n concurrent launches: same results, n < 1 refused."""
    from nav2_social_mpc_controller_amd.solver import BatchSolver, SmpcError

    prm = OptimizerParams.readme()
    sc = make_scenes(prm, 8192 + 37, 8)   # more scene groups than the shared grid has waves: several scenes per slot
    s = BatchSolver(prm)
    base = s.solve(sc)
    for n in (3, 12, 1000):
        s.set_solve_share(n)
        got = s.solve(sc)
        for k in base:
            assert np.array_equal(base[k], got[k]), (n, k)
    with pytest.raises(SmpcError):
        s.set_solve_share(0)


@pytest.mark.gpu
def test_solve_slot_width_rule():
    """smpc_solve_slot_width: one scene per wave for every shape beyond 31 steps / 32 agents, and for the shapes that fit
    two per wave while the batch is small (one scene per SIMD; eight waves per CU where helper lanes pay); a function of
    the batch's shape and the handle's share alone."""
    from nav2_social_mpc_controller_amd.solver import BatchSolver, SmpcError

    s = BatchSolver(OptimizerParams.readme())
    cus = 256  # MI355X
    assert s.solve_slot_width(8192, 38, 16) == 64 and s.solve_slot_width(1, 38, 3) == 64
    assert s.solve_slot_width(8192, 28, 64) == 64
    assert s.solve_slot_width(8192, 28, 8) == 32
    assert s.solve_slot_width(1, 28, 8) == 64 and s.solve_slot_width(8 * cus, 28, 8) == 64
    assert s.solve_slot_width(8 * cus + 1, 28, 8) == 32
    assert s.solve_slot_width(4 * cus, 28, 3) == 64 and s.solve_slot_width(4 * cus + 1, 28, 3) == 32
    assert s.solve_slot_width(4 * cus, 28, 0) == 64
    s.set_solve_share(3)
    assert s.solve_slot_width(8 * cus // 3, 28, 8) == 64 and s.solve_slot_width(8 * cus // 3 + 1, 28, 8) == 32
    s.set_solve_share(1)
    with pytest.raises(SmpcError):
        s.solve_slot_width(1, 64, 3)


@pytest.mark.gpu
@pytest.mark.parametrize("B,N,which", [(300, 8, "readme"), (4400, 8, "readme"), (160, 3, "params_yaml"), (200, 0, "readme")])
def test_stopping_at_the_grams_last_column_changes_nothing(B, N, which, monkeypatch):
    """A sweep forms the whole Gram only where the point can be adopted (initial point, Armijo passed, re-evaluation);
    a line-search sample that fails the Armijo test stops at the last column (cost, gradient). Every entry is summed in
    the same order either way, so with SMPC_FULL_GRAM=1 (every sweep forms everything, read at every launch) the
    results are the same bit for bit: one scene per wave with helper lanes (B = 300), two scenes per wave (4400), the
    reference's params.yaml shape (P = 10), a batch without people."""
    from nav2_social_mpc_controller_amd.solver import BatchSolver

    prm = OptimizerParams.readme() if which == "readme" else OptimizerParams.params_yaml()
    sc = make_scenes(prm, B, max(N, 1), people_present=N > 0)
    s = BatchSolver(prm)
    monkeypatch.delenv("SMPC_FULL_GRAM", raising=False)
    lazy = s.solve(sc)
    monkeypatch.setenv("SMPC_FULL_GRAM", "1")
    full = s.solve(sc)
    assert lazy["evaluations"].sum() > 3 * lazy["iterations"].sum() / 2   # there are line-search samples to skip on
    for k in lazy:
        assert np.array_equal(lazy[k], full[k], equal_nan=True), k
