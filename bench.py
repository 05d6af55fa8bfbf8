#!/usr/bin/env python3
"""bench.py — MPC solves/sec of the batched social-MPC hot path on N MI355X GPUs of one node.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N>1 it is launched through
`python -m torch.distributed.run --nproc-per-node N ...` (one rank per GPU, RCCL). One "step" = one batched solve
(smpc_solve_batch) of the per-GPU batch of synthetic crowd scenes with all inputs already resident in HBM.
Weak scaling: every rank owns its own 8192 scenes (BASELINE.json configs[2] per GPU; configs[3] = 8 GPUs x 8192),
regenerated from (seed, scene_id) — the path shards with no data-path collective.
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def usable_cores():
    """Host cores this process may really use: affinity mask, capped by the cgroup CPU quota when there is one."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // per))
        except Exception:
            pass
    return max(1, min(n, 64))


def pmc_traffic_bytes(kernel="smpc_solve_kernel"):
    """HBM bytes per launch of `kernel` from the newest committed PMC summary (profiles/rNN_pmc_summary.txt, made by
    tools/profile_round.sh with separate --pmc passes): (FETCH_SIZE + WRITE_SIZE) KiB * 1024. The gfx950 x2
    correction of FETCH_SIZE applies to 16 B/lane streaming reads only; this kernel reads 8 B/lane and byte gathers,
    for which the counter is uncalibrated (MI355X_MICROARCH.md, HBM) — no correction applied. None if unavailable."""
    import glob
    import re
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_summary.txt")))
    if not files:
        return None, None
    fetch = write = None
    active = False
    for line in open(files[-1]):
        if line.startswith("=="):
            active = kernel in line
        elif active:
            m = re.match(r"\s*(FETCH_SIZE|WRITE_SIZE)\s+mean/dispatch\s+([0-9.e+]+)", line)
            if m:
                if m.group(1) == "FETCH_SIZE":
                    fetch = float(m.group(2))
                else:
                    write = float(m.group(2))
    if fetch is None or write is None:
        return None, None
    return (fetch + write) * 1024.0, os.path.basename(files[-1])


def algorithmic_bytes_per_sweep(N, T, P, M):
    """every input read once, J and r written once, per scene per sweep."""
    return 8 * (6 * N * T + 2 * (T + 1) + P + 5) + 16 * T + 8 * (M * P + M)


def closed_loop_extras(prm, scenes, device_index, ticks=10):
    """SURVEY §8 rows f1-f3 measured beside the solve: receding-horizon ticks (trajectorize -> format_to_optimize ->
    project_people -> solve -> memory store), everything resident in HBM, plus the HIP-event time of each stage."""
    import numpy as np

    from nav2_social_mpc_controller_amd.episode import BatchEpisode, arc_plans
    from nav2_social_mpc_controller_amd.params import TrajectorizerParams
    from nav2_social_mpc_controller_amd.scenes import uniform

    B, T, N = scenes.B, scenes.T, scenes.N
    tp = TrajectorizerParams(desired_linear_vel=0.6, lookahead_dist=0.4, max_angular_vel=1.0, time_step=0.05,
                             max_time=float(prm.max_time))
    curv = (uniform(0x5EED0001, np.arange(B), 6)[:, 0] * 2.0 - 1.0) * 0.24
    L = 400
    plan, plan_len = arc_plans(scenes.pose0, curv, L=L)
    ep = BatchEpisode(prm, scenes, curv, np.zeros((480, 480), np.uint32), np.array([-16.0, -16.0]),
                      float(np.float32(0.1)), device=device_index, plan=plan, plan_len=plan_len, traj_params=tp,
                      fov_angle=np.pi)  # pi: everybody on the costmap is seen, the 8-agent workload of the headline config
    for _ in range(2):
        ep.tick()
    ep.synchronize()
    t0 = time.perf_counter()
    for _ in range(ticks):
        ep.tick()
    ep.synchronize()
    tick_s = (time.perf_counter() - t0) / ticks
    tm = {}
    ep.tick(timing=tm)
    S1 = tp.max_steps + 1
    alg = {  # algorithmic bytes per scene: every input read once, every output written once
        "trajectorize": 8 * (2 * L + 3) + 4 + 8 * S1 * (3 + 2 + 1) + 8,
        "format": 8 * (T + 1) * (3 + 2 + 3 + 2) + 16 + 8 * ((T + 1) * (6 + 2) + 3 + 6 + 1),
        "project": 8 * (6 * N + 6 * (T + 1)) + 8 * 6 * N * (T + 1) + 4,
        "store": 8 * (T + 1) * 5 * 2 + 8,
    }
    alg["people"] = 8 * 5 * N + 4 + 8 * 3 + 8 * 6 * N + 1
    stages = {}
    for k in ("trajectorize", "people", "format", "project", "store"):
        ms = tm[k + "_ms"]
        stages[k] = {"kernel_ms": ms, "algorithmic_GBps": B * alg[k] / (ms * 1e-3) / 1e9, "bytes_per_scene": alg[k]}
    stages["solve"] = {"kernel_ms": tm["solve_ms"]}
    return {"ticks_per_s": B / tick_s, "ms_per_tick": tick_s * 1e3, "ticks_timed": ticks,
            "chain": "trajectorize(f3) -> fov filter + people_to_status(f4, f2) -> format_to_optimize(f2) -> project_people(f1) "
                     "-> solve(a1-a12) -> memory store(f2)",
            "stages": stages, "last_tick_failures": int((ep.res["status"] == 2).sum().item()),
            "projection_errors": int((ep.proj_error != 0).sum().item())}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=8192, help="scenes per GPU")
    ap.add_argument("--people", type=int, default=8)
    ap.add_argument("--fixed-iterations", type=int, default=0, help="1: run exactly 40 LM iterations per scene")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the extra measurements in `config` (PCIe-inclusive, fixed-40, closed loop): profiling runs")
    ap.add_argument("--streams", type=int, default=4,
                    help="consecutive steps are issued round-robin on this many HIP streams (one solver handle each)")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    from nav2_social_mpc_controller_amd import dist as D
    from nav2_social_mpc_controller_amd.params import OptimizerParams
    from nav2_social_mpc_controller_amd.scenes import make_scenes
    from nav2_social_mpc_controller_amd.solver import BatchSolver

    rank, local_rank, world = D.env_rank_world()
    force_dist = os.environ.get("SMPC_BENCH_FORCE_DIST") == "1"   # rehearse the RCCL path with a single rank
    if args.gpus > 1 or world > 1 or force_dist:
        assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run"
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        try:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        except TypeError:  # older torch: no device_id argument
            dist.init_process_group("nccl")
    device = torch.device("cuda", local_rank)
    torch.cuda.set_device(device)

    prm = OptimizerParams.readme().replace(fixed_iterations=args.fixed_iterations)
    B, N = args.batch, args.people
    lo, hi = D.weak_shard(B, rank)
    scenes = make_scenes(prm, B, N, seed=0x5EED0001, first_scene=lo)
    T = scenes.T
    CH, bl, nb, P, M, _ = prm.dims(T, True)

    # Consecutive steps (independent batches of a serving stream) are issued round-robin on a few HIP streams, one
    # solver handle each: the persistent solve kernel ends with a tail of a few long scenes (per-scene LM iteration
    # counts vary 5..40), and the next batch's waves fill the CUs that the tail leaves idle. Every step still solves
    # the whole batch; results of step k land in result set k % streams.
    n_streams = max(1, args.streams)
    solvers = [BatchSolver(prm, device=local_rank) for _ in range(n_streams)]
    hip_streams = [torch.cuda.Stream(device=device) for _ in range(n_streams)]
    for sv, st in zip(solvers, hip_streams):
        sv.set_stream(st.cuda_stream)
    solver = solvers[0]
    sb, tens = scenes.to_device(device)
    results = [sv.alloc_results(B, T, device) for sv in solvers]
    rb, out = results[0]
    eo, eout = solver.alloc_eval(B, T, device)

    def barrier():
        torch.cuda.synchronize(device)
        if dist.is_initialized():
            dist.barrier()
        torch.cuda.synchronize(device)

    for w in range(max(args.warmup, n_streams)):
        solvers[w % n_streams].solve_device(sb, results[w % n_streams][0])
    barrier()
    t0 = time.perf_counter()
    for k in range(args.steps):
        solvers[k % n_streams].solve_device(sb, results[k % n_streams][0])
    barrier()
    elapsed = time.perf_counter() - t0
    # per-launch device time of the solve kernel: HIP events around each handle's last launch, on its own stream
    solve_ms_each = [sv.last_kernel_ms() for sv in solvers[:min(n_streams, args.steps)]]
    solve_ms = float(sum(solve_ms_each) / len(solve_ms_each))
    # an un-overlapped launch for reference (single stream, nothing else in flight)
    solver.solve_device(sb, rb)
    torch.cuda.synchronize(device)
    solo_ms = solver.last_kernel_ms()

    tmax = torch.tensor([elapsed], dtype=torch.float64, device=device)
    if dist.is_initialized():
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    elapsed_max = float(tmax.item())

    evals = out["evaluations"].cpu().numpy().astype(np.int64)
    iters = out["iterations"].cpu().numpy()
    status = out["status"].cpu().numpy()
    local = {"scenes": B, "sweeps": int(evals.sum()), "iterations": int(iters.sum()),
             "converged": int((status == 0).sum()), "no_convergence": int((status == 1).sum()),
             "failed": int((status == 2).sum()), "max_solve_kernel_ms": solve_ms, "max_solo_kernel_ms": solo_ms}
    summ = D.reduce_summary(local, device=device)

    # K1 stand-alone sweep (residual + Jacobian rows written to HBM) for the roofline line
    for _ in range(3):
        solver.eval_device(sb, tens["init_params"].data_ptr(), eo)
    k1_ms = solver.last_kernel_ms()

    if rank == 0:
        total_solves = summ["scenes"] * args.steps
        value = total_solves / elapsed_max
        bytes_sweep = algorithmic_bytes_per_sweep(N, T, P, M)
        sweeps_per_launch = int(evals.sum())
        # roofline of the solve kernel: one launch alone on the GPU (HIP events on its own stream, taken right after the
        # timed region) — inside the timed region launches of consecutive steps overlap, their individual durations
        # are not a per-kernel efficiency. The aggregate rate over the timed region is reported next to it.
        achieved = sweeps_per_launch * bytes_sweep / (solo_ms * 1e-3) / 1e9
        aggregate = args.steps * sweeps_per_launch * bytes_sweep / elapsed_max / 1e9
        k1_achieved = B * bytes_sweep / (k1_ms * 1e-3) / 1e9
        traffic, traffic_src = pmc_traffic_bytes()
        line = {
            "metric": "MPC solves/sec (horizon=18, 8 agents, 40 iters)", "value": value, "unit": "solves/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed_max / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"BASELINE configs[2] per GPU: batch={B} scenes/GPU, {N} people, horizon=18 "
                                   f"(T={T}, block=6, P={P}, M={M}), 200x200 u8 costmap per scene, DENSE_SCHUR, "
                                   f"max 40 LM iterations with Ceres termination rules"
                                   + (" DISABLED (fixed 40 iterations)" if args.fixed_iterations else ""),
                       "scenes_per_gpu": B, "people": N, "T": T, "P": P, "M": M, "streams": n_streams,
                       "single_stream_solves_per_s_per_gpu": B / (summ["max_solo_kernel_ms"] * 1e-3),
                       "mean_lm_iterations": summ["iterations"] / summ["scenes"],
                       "mean_sweeps_per_solve": summ["sweeps"] / summ["scenes"],
                       "status": {"convergence": int(summ["converged"]), "no_convergence": int(summ["no_convergence"]),
                                  "failure": int(summ["failed"])}},
            "roofline": {"bound": "hbm", "kernel": "smpc_solve_kernel", "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "traffic_source": traffic_src, "algorithmic_bytes_per_launch": sweeps_per_launch * bytes_sweep,
                         "launch_ms": solo_ms, "launch_ms_overlapped_in_timed_region": solve_ms,
                         "launch_note": "launch_ms = HIP-event duration of one solve launch alone on the GPU (what "
                                        "`bench.py --streams 1` and the committed rocprofv3 kernel stats show); in the timed "
                                        "region launches of consecutive steps overlap on %d streams" % n_streams,
                         "aggregate_achieved_in_timed_region": aggregate, "aggregate_frac": aggregate / HBM_PEAK_GBS,
                         "sweeps_per_launch": sweeps_per_launch, "bytes_per_sweep": bytes_sweep,
                         "k1_sweep_kernel": {"launch_ms": k1_ms, "achieved": k1_achieved, "frac": k1_achieved / HBM_PEAK_GBS}},
        }
        if world == 1 and not args.no_extras:
            # extras of SURVEY §8(d): (ii) end-to-end including PCIe staging, (iii) fixed-40-iteration mode
            t1 = time.perf_counter()
            solver.solve(scenes)                      # host pointers: H2D of every input, solve, D2H of every output
            line["config"]["pcie_inclusive_solves_per_s"] = B / (time.perf_counter() - t1)
            fixed = BatchSolver(prm.replace(fixed_iterations=1), device=local_rank)
            frb, fout = fixed.alloc_results(B, T, device)
            fixed.solve_device(sb, frb)
            fixed.solve_device(sb, frb)
            torch.cuda.synchronize(device)
            fms = fixed.last_kernel_ms()
            line["config"]["fixed_40_iterations"] = {"launch_ms": fms, "solves_per_s": B / (fms * 1e-3),
                                                     "mean_sweeps_per_solve": float(fout["evaluations"].float().mean().item())}
        if world == 1 and not args.no_extras:
            line["config"]["closed_loop"] = closed_loop_extras(prm, scenes, local_rank)
        if world == 1 and not args.no_cpu_baseline:
            from oracle import oracle_py as O
            cores = usable_cores()
            n_sample = min(B, 512 * cores)   # ~10-30 s of CPU work at ~35 solves/s/core
            sample = scenes.select(np.arange(n_sample))
            t1 = time.perf_counter()
            ref = O.solve(prm, sample, nthreads=cores)          # reference-literal oracle: the timed CPU baseline
            cpu_s = time.perf_counter() - t1
            refz = O.solve(prm, sample, nthreads=cores, theta_zero_convention=True)   # checker
            got = out["cmds"][:n_sample].cpu().numpy()
            dcmd = np.abs(got - refz["cmds"]).reshape(n_sample, -1).max(axis=1)
            firm = refz["marginal_decisions"] == 0
            clean = (ref["sign_noise_events"] == 0) & (ref["marginal_decisions"] == 0)
            dlit = np.abs(got - ref["cmds"]).reshape(n_sample, -1).max(axis=1)
            line["cpu_baseline"] = {"value": n_sample / cpu_s, "unit": "solves/s", "cores": cores, "kind": "port",
                                    "sample": f"first {n_sample} scenes of the same workload, one solve per thread "
                                              f"(CPU restatement oracle/smpc_oracle.cpp, not Ceres), {cpu_s:.1f} s"}
            line["parity"] = {"scenes": int(n_sample), "scenes_with_firm_decisions": int(firm.sum()),
                              "max_abs_dcmd": float(dcmd[firm].max()), "median_abs_dcmd": float(np.median(dcmd)),
                              "scenes_over_1e-5": int((dcmd[firm] > 1e-5).sum()),
                              "scenes_over_1e-5_among_marginal": int((dcmd[~firm] > 1e-5).sum()),
                              "checker": "CPU oracle, theta:=0 convention for exactly equal velocities",
                              "literal_oracle": {"scenes_without_sign_noise": int(clean.sum()),
                                                 "max_abs_dcmd_on_those": float(dlit[clean].max()) if clean.any() else None}}
        print(json.dumps(line))
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
