"""Development perf probe: K1 sweep and fused solve kernel timing on device-resident batches."""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from nav2_social_mpc_controller_amd.params import OptimizerParams
from nav2_social_mpc_controller_amd.scenes import make_scenes
from nav2_social_mpc_controller_amd.solver import BatchSolver

def bench(name, p, B, N, reps=3, **kw):
    sc = make_scenes(p, B, N, **kw)
    CH, bl, nb, P, M, _ = p.dims(sc.T, True)
    s = BatchSolver(p)
    sb, tens = sc.to_device()
    rb, rt = s.alloc_results(B, sc.T)
    eo, et = s.alloc_eval(B, sc.T)
    torch.cuda.synchronize()
    for i in range(reps):
        s.eval_device(sb, tens["init_params"].data_ptr(), eo)
        ms_raw = s.last_kernel_ms()
    for i in range(reps):
        s.solve_device(sb, rb)
        ms_solve_raw = s.last_kernel_ms()
    keep = s.stage_people_device(sb)          # from here on the batch carries its staged people block
    stage_ms = s.last_kernel_ms()
    for i in range(reps + 2):
        s.eval_device(sb, tens["init_params"].data_ptr(), eo)
        ms = s.last_kernel_ms()
    print(f"[{name}] staging pass {stage_ms*1e3:.1f} us; K1 kernel with library-side staging before it {ms_raw*1e3:.1f} us; solve kernel (raw people input) {ms_solve_raw:.3f} ms")
    bytes_sweep = 8 * (6 * N * sc.T + 2 * (sc.T + 1) + P + 5) + 16 * sc.T + 8 * (M * P + M)
    print(f"[{name}] K1 eval B={B}: {ms:.3f} ms -> {B/ms*1e3:.3e} scene-sweeps/s, {B*bytes_sweep/ms/1e6:.1f} GB/s algorithmic ({bytes_sweep} B/sweep)")
    for i in range(reps):
        s.solve_device(sb, rb)
        ms = s.last_kernel_ms()
    ev = rt["evaluations"].cpu().numpy(); it = rt["iterations"].cpu().numpy()
    print(f"[{name}] solve B={B}: {ms:.3f} ms -> {B/ms*1e3:.1f} solves/s; evals mean {ev.mean():.1f} max {ev.max()} iters mean {it.mean():.1f}; {ms*1e3/ev.sum()*1e3:.2f} ns per scene-sweep; status {np.bincount(rt['status'].cpu().numpy(), minlength=3)}")

if __name__ == "__main__":
    p = OptimizerParams.readme()
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
    bench("cfg3-N8", p, B, 8)
    bench("cfg3-N8-fixed40", p.replace(fixed_iterations=1), B, 8)
    bench("cfg2-N4", p, 1024, 4)
    bench("nopeople", p, B, 3, people_present=False)
