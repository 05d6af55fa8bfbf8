"""Development probe: the serial floor of a lone solve launch — the scenes with the most sweeps of the headline batch,
solved alone (every wave has the GPU to itself), next to the full batch."""
import sys
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
for _ in range(3):
    s.solve_device(sb, rb); torch.cuda.synchronize()
full = s.last_kernel_ms()
ev = rt["evaluations"].cpu().numpy(); it = rt["iterations"].cpu().numpy()
order = np.argsort(-ev)
print(f"full batch {full:.3f} ms; sweeps mean {ev.mean():.1f} p50 {np.percentile(ev,50):.0f} p90 {np.percentile(ev,90):.0f} p99 {np.percentile(ev,99):.0f} max {ev.max()}; iterations max {it.max()}")
for k in (2, 64, 512, 2048):
    sub = sc.select(np.sort(order[:k]))
    sb2, t2 = sub.to_device()
    rb2, rt2 = s.alloc_results(k, sc.T)
    ms = []
    for _ in range(3):
        s.solve_device(sb2, rb2); torch.cuda.synchronize(); ms.append(s.last_kernel_ms())
    e2 = rt2["evaluations"].cpu().numpy()
    print(f"  top {k:5d} scenes alone: {min(ms):.3f} ms (sweeps {e2.min()}..{e2.max()}) -> {min(ms)*1e3/e2.max():.1f} us per sweep of the longest scene")
sub = sc.select(np.sort(order[-2048:]))
sb2, t2 = sub.to_device(); rb2, rt2 = s.alloc_results(2048, sc.T)
for _ in range(3):
    s.solve_device(sb2, rb2); torch.cuda.synchronize()
print(f"  cheapest 2048 alone: {s.last_kernel_ms():.3f} ms (sweeps max {rt2['evaluations'].max().item()})")
# longest-first (perfect knowledge) and a noisy predictor: the batch physically permuted
rng = np.random.default_rng(1)
for name, key in (("perfect LPT", -ev.astype(float)), ("LPT with 20 % noise", -ev * (1 + 0.2 * rng.standard_normal(B))),
                  ("LPT on 8-sweep buckets", -(ev // 8).astype(float)), ("shortest first", ev.astype(float))):
    sub = sc.select(np.argsort(key, kind="stable"))
    sb2, t2 = sub.to_device(); rb2, rt2 = s.alloc_results(B, sc.T)
    ms = []
    for _ in range(3):
        s.solve_device(sb2, rb2); torch.cuda.synchronize(); ms.append(s.last_kernel_ms())
    print(f"  {name}: {min(ms):.3f} ms")
# the same through the queue-order hint (smpc_scene_batch.order), scenes left where they are
for name, key in (("order hint, perfect", -ev.astype(float)), ("order hint, 8-sweep buckets", -(ev // 8).astype(float))):
    order = torch.from_numpy(np.argsort(key, kind="stable").astype(np.int32)).to("cuda:0")
    sb.order = order.data_ptr()
    ms = []
    for _ in range(3):
        s.solve_device(sb, rb); torch.cuda.synchronize(); ms.append(s.last_kernel_ms())
    print(f"  {name}: {min(ms):.3f} ms")
sb.order = None
