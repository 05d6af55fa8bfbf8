"""The N>1 bench path on CPU: two gloo ranks each own a shard of scenes (regenerated from scene ids, no data-path
collective) and reduce only summary scalars, exactly as bench.py does over RCCL. The solve itself has no CPU
path, so each rank stands in the oracle for its shard here — the point is the sharding / reduction logic."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, total, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from nav2_social_mpc_controller_amd import dist as D
    from nav2_social_mpc_controller_amd.params import OptimizerParams
    from nav2_social_mpc_controller_amd.scenes import make_scenes
    from oracle import oracle_py as O
    p = OptimizerParams.readme()
    lo, hi = D.shard_range(total, rank, world)
    sc = make_scenes(p, hi - lo, 3, map_cells=40, seed=5, first_scene=lo)
    res = O.solve(p, sc)
    dist.barrier()
    summ = D.reduce_summary({"scenes": hi - lo, "iterations": int(res["iterations"].sum()),
                             "max_seconds": 0.5 + rank, "cmd_sum": float(res["cmds"].sum()),
                             # the bench's per-rank parity record: MAX over ranks of max |dcmd|, SUM of the counts
                             "max_abs_dcmd": 1e-9 * (rank + 1), "scenes_over_1e-5": rank})
    gathered = D.gather_params(torch.from_numpy(res["params"]))      # [world][B/world][P] on every rank
    q.put((rank, lo, hi, summ, res["cmds"], gathered.numpy()))
    dist.destroy_process_group()


def test_two_rank_weak_sharding_matches_single_process():
    total, world = 6, 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, total, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted([q.get(timeout=240) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    sys.path.insert(0, ROOT)
    from nav2_social_mpc_controller_amd.params import OptimizerParams
    from nav2_social_mpc_controller_amd.scenes import make_scenes
    from oracle import oracle_py as O
    prm = OptimizerParams.readme()
    whole = O.solve(prm, make_scenes(prm, total, 3, map_cells=40, seed=5))
    cmds = np.concatenate([g[4] for g in got], axis=0)
    assert np.array_equal(cmds, whole["cmds"])                      # shards == the unsharded batch, bit for bit
    for _, lo, hi, summ, _, gathered in got:
        assert gathered.shape == (world, total // world, whole["params"].shape[1])
        assert np.array_equal(gathered.reshape(total, -1), whole["params"])   # all_gather: every rank holds every shard
        assert summ["max_abs_dcmd"] == 2e-9 and summ["scenes_over_1e-5"] == 1   # MAX / SUM of the parity record
        assert summ["scenes"] == total                              # SUM
        assert summ["iterations"] == int(whole["iterations"].sum())
        assert summ["max_seconds"] == 1.5                           # MAX over ranks
        assert abs(summ["cmd_sum"] - float(whole["cmds"].sum())) < 1e-9
