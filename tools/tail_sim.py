"""Development probe: how much of the solve kernel's makespan is tail imbalance? Greedy queue simulation of the
per-scene sweep counts on 4096 slots (2048 waves x 2), with slot pairs tied to one wave (a wave runs while either slot is busy)."""
import sys, heapq
import numpy as np, torch
sys.path.insert(0, ".")
from nav2_social_mpc_controller_amd.params import OptimizerParams
from nav2_social_mpc_controller_amd.scenes import make_scenes
from nav2_social_mpc_controller_amd.solver import BatchSolver
p = OptimizerParams.readme()
B = 8192
sc = make_scenes(p, B, 8)
s = BatchSolver(p)
sb, tens = sc.to_device()
rb, rt = s.alloc_results(B, sc.T)
s.solve_device(sb, rb); s.solve_device(sb, rb)
ms = s.last_kernel_ms()
ev = rt["evaluations"].cpu().numpy().astype(np.int64)
print("kernel ms", ms, "sweeps total", ev.sum(), "mean", ev.mean(), "max", ev.max(), "p99", np.percentile(ev, 99))
for nslots in (4096,):
    # slots pull scenes in index order; time unit = one sweep (a wave advances both of its slots together)
    free = [(0, i) for i in range(nslots)]
    heapq.heapify(free)
    end = np.zeros(nslots)
    for e in ev:
        t, i = heapq.heappop(free)
        heapq.heappush(free, (t + e, i))
        end[i] = t + e
    makespan = end.max()
    wave_end = np.maximum(end[0::2], end[1::2])
    print(f"slots {nslots}: ideal {ev.sum()/nslots:.1f} sweeps, makespan {makespan:.0f} sweeps -> efficiency {ev.sum()/nslots/makespan:.3f}; "
          f"mean wave end {wave_end.mean():.1f}")
    print(f"  => per-wave-sweep time {ms*1e3/makespan:.2f} us; a perfectly balanced run would take {ms*ev.sum()/nslots/makespan:.3f} ms")

def makespan(order, nslots=4096):
    free = [(0, i) for i in range(nslots)]
    heapq.heapify(free)
    end = 0
    for e in ev[order]:
        t, i = heapq.heappop(free)
        heapq.heappush(free, (t + e, i))
        end = max(end, t + e)
    return end
ic = rt["initial_cost"].cpu().numpy()
from scipy.stats import spearmanr
print("spearman(initial_cost, sweeps) =", spearmanr(ic, ev).correlation)
ppl = sc.people
d0 = np.sqrt((ppl[:, 1, 0, :] - sc.pose0[:, None, 0]) ** 2 + (ppl[:, 1, 1, :] - sc.pose0[:, None, 1]) ** 2).min(axis=1)
print("spearman(min agent distance, sweeps) =", spearmanr(d0, ev).correlation)
print("makespan index order", makespan(np.arange(B)), " by initial cost desc", makespan(np.argsort(-ic)), " perfect LPT", makespan(np.argsort(-ev)), " ideal", ev.sum() / 4096)
