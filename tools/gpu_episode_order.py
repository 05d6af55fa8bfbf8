import sys, numpy as np, torch
sys.path.insert(0, ".")
from nav2_social_mpc_controller_amd.episode import BatchEpisode, arc_plans
from nav2_social_mpc_controller_amd.params import OptimizerParams, TrajectorizerParams
from nav2_social_mpc_controller_amd.scenes import make_scenes, uniform
from scipy.stats import spearmanr
prm = OptimizerParams.readme(); B, N = 8192, 8
sc = make_scenes(prm, B, N)
w_ref = (uniform(0x5EED0001, np.arange(B), 6)[:, 0] * 2.0 - 1.0) * 0.6
plan, plan_len = arc_plans(sc.pose0, 0.4 * w_ref)
kw = dict(plan=plan, plan_len=plan_len, traj_params=TrajectorizerParams(desired_linear_vel=0.6, max_time=prm.max_time))
for hint in (False, True):
    ep = BatchEpisode(prm, sc, w_ref, np.zeros((480, 480), np.uint32), np.array([-16.0, -16.0]), 0.1, order_hint=hint, **kw)
    prev = None
    for t in range(8):
        tm = {}
        ep.tick(timing=tm)
        ev = ep.res["evaluations"].cpu().numpy()
        c = spearmanr(prev, ev).correlation if prev is not None else float("nan")
        print(f"hint {hint} tick {t}: solve {tm['solve_ms']:.3f} ms, sweeps mean {ev.mean():.1f} max {ev.max()}, spearman with previous tick {c:.3f}")
        prev = ev
