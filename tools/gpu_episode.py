"""Development probe: closed-loop batch episode throughput (format + project + solve + store per tick).
usage: gpu_episode.py [B] [N] [ticks] [plan|window] [shard counts...]: a negative shard count replays HIP graphs, `0` skips the
sharded runs (single chain + stage times only), no count runs -2 -3 -4 and then the single chain."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from nav2_social_mpc_controller_amd.episode import BatchEpisode, ShardedEpisode, arc_plans
from nav2_social_mpc_controller_amd.params import OptimizerParams, TrajectorizerParams
from nav2_social_mpc_controller_amd.scenes import make_scenes, uniform

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
N = int(sys.argv[2]) if len(sys.argv) > 2 else 8
ticks = int(sys.argv[3]) if len(sys.argv) > 3 else 20
prm = OptimizerParams.readme()
sc = make_scenes(prm, B, N)
w_ref = (uniform(0x5EED0001, np.arange(B), 6)[:, 0] * 2.0 - 1.0) * 0.6
kw = {}
if len(sys.argv) > 4 and sys.argv[4] in ("plan", "window"):  # global plans trajectorized on the device every tick (row f3)
    plan, plan_len = arc_plans(sc.pose0, 0.4 * w_ref)
    kw = dict(plan=plan, plan_len=plan_len, traj_params=TrajectorizerParams(desired_linear_vel=0.6, max_time=prm.max_time))
    if sys.argv[4] == "window":  # with PathHandler::transformGlobalPlan in front (row f4): 10 m = half the 20 m costmap
        kw["plan_window"] = (10.0, 10.0)
ep = BatchEpisode(prm, sc, w_ref, np.zeros((480, 480), np.uint32), np.array([-16.0, -16.0]), 0.1, **kw)
for _ in range(2):
    ep.tick()
ep.synchronize()
t0 = time.perf_counter()
for _ in range(ticks):
    ep.tick()
ep.synchronize()
dt = (time.perf_counter() - t0) / ticks
only = [int(a) for a in sys.argv[5:]]
for shards in ([] if only == [0] else (only or (-2, -3, -4))):  # the same robots as independent chains on separate streams
    se = ShardedEpisode(prm, sc, w_ref, np.zeros((480, 480), np.uint32), np.array([-16.0, -16.0]), 0.1, shards=abs(shards), graphs=shards < 0,
                        solve_share=int(os.environ["SMPC_SHARE"]) if "SMPC_SHARE" in os.environ else None, **kw)
    for _ in range(2):
        se.tick()
    se.synchronize()
    t0 = time.perf_counter()
    for _ in range(ticks):
        se.tick()
    se.synchronize()
    ds = (time.perf_counter() - t0) / ticks
    tag = " (HIP graphs)" if shards < 0 else ""
    print(f"  {abs(shards)} shards{tag}: {ds*1e3:.3f} ms/tick -> {B/ds:.0f} controller ticks/s")
    del se
if only and only != [0]:
    sys.exit(0)
tm = {}
acc = {}
for _ in range(5):  # per-stage HIP-event times (each stage synchronised: not the pipelined tick time)
    ep.tick(timing=tm)
    for k, v in tm.items():
        acc[k] = acc.get(k, 0.0) + v / 5
print("stage ms:", {k: round(v, 4) for k, v in acc.items()})
st = ep.res["status"].cpu().numpy()
print(f"episode B={B} N={N}: {dt*1e3:.3f} ms/tick -> {B/dt:.0f} controller ticks/s; last-tick status {np.bincount(st, minlength=3)} "
      f"iters mean {ep.res['iterations'].double().mean().item():.1f} sweeps mean {ep.res['evaluations'].double().mean().item():.1f}; proj errors {(ep.proj_error != 0).sum().item()}")
