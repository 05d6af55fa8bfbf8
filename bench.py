#!/usr/bin/env python3
"""bench.py — MPC solves/sec of the batched social-MPC hot path on N MI355X GPUs of one node.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N>1 it is launched through
`python -m torch.distributed.run --nproc-per-node N ...` (one rank per GPU, RCCL). One "step" = one batched solve
(smpc_solve_batch) of the per-GPU batch of synthetic crowd scenes with all inputs already resident in HBM in the
reference's layout (the people block is staged by the library inside the step).
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

HBM_PEAK_GBS = 8000.0    # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP64_PEAK_TFS = 78.6     # public datasheet FP64 vector peak; bench measures the real one beside it (smpc_fp64_peak_probe)
SIMDS = 1024             # 256 CUs x 4 SIMDs
VALU_CYCLES = 4          # cycles one wave64 VALU instruction occupies its SIMD (measured: SQ_ACTIVE_INST_VALU x 4 / SQ_INSTS_VALU)


def launch_ranks(n, argv):
    """`python bench.py --gpus N` without a torchrun environment: start N copies of this script, one rank per GPU
    (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT as torch.distributed.run would set them), relay rank 0's
    JSON line, and fail if any rank fails. Runs before anything touches the GPU: this process never initialises HIP."""
    import socket
    import subprocess

    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), SMPC_BENCH_SPAWNED="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # RCCL needs dmabuf IPC on this driver
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, text=True))
    # rank 0's stdout is read on a thread while every rank is polled: a rank that dies leaves the others waiting in a
    # collective, so the first non-zero exit ends the run (the remaining ranks — exactly the PIDs started here — are
    # terminated) and no line is printed
    import threading
    buf = []
    reader = threading.Thread(target=lambda: buf.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    codes = [None] * n
    while any(c is None for c in codes):
        for r, pr in enumerate(procs):
            if codes[r] is None:
                codes[r] = pr.poll()
        if any(c not in (None, 0) for c in codes):
            for r, pr in enumerate(procs):
                if codes[r] is None:
                    pr.terminate()
            for r, pr in enumerate(procs):
                if codes[r] is None:
                    try:
                        codes[r] = pr.wait(timeout=20)
                    except subprocess.TimeoutExpired:
                        pr.kill()
                        codes[r] = pr.wait()
            break
        time.sleep(0.05)
    reader.join(timeout=10)
    if any(c != 0 for c in codes):
        sys.stderr.write(f"bench.py: rank exit codes {codes}\n")
        return next(c for c in codes if c not in (0, None)) or 1
    sys.stdout.write("".join(buf))
    sys.stdout.flush()
    return 0


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


def pmc_counters(kernel="smpc_solve_kernel"):
    """Per-launch PMC counters of `kernel` from the newest committed summary (profiles/rNN_pmc_summary.txt, made by
    tools/profile_round.sh with separate --pmc passes) — but only if that summary was taken on the device sources this
    process runs (`# csrc_digest:` line == buildinfo.csrc_digest()). Returns (counters, info); counters is {} when there
    is no summary or it belongs to another build, and info says which."""
    import glob
    import re

    from nav2_social_mpc_controller_amd.buildinfo import csrc_digest
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_summary.txt")))
    here = csrc_digest()
    if not files:
        return {}, {"file": None, "csrc_digest_running": here, "usable": False, "why": "no profiles/r*_pmc_summary.txt"}
    out = {}
    active = False
    digest = None
    for line in open(files[-1]):
        if line.startswith("# csrc_digest:"):
            digest = line.split(":", 1)[1].strip()
        elif line.startswith("=="):
            active = kernel in line
        elif active:
            m = re.match(r"\s*(\w+)\s+mean/dispatch\s+([0-9.e+]+)", line)
            if m:
                out[m.group(1)] = float(m.group(2))
    info = {"file": os.path.basename(files[-1]), "csrc_digest_of_profile": digest, "csrc_digest_running": here,
            "usable": digest == here}
    if digest != here:
        info["why"] = "the committed PMC summary was taken on other device sources: its instruction counts are not used"
        return {}, info
    return out, info


def algorithmic_bytes_per_sweep(N, T, P, M):
    """every input read once, J and r written once, per scene per sweep."""
    return 8 * (6 * N * T + 2 * (T + 1) + P + 5) + 16 * T + 8 * (M * P + M)


# PathHandler::transformGlobalPlan as the reference calls it: 4 m of path length for the closest-pose search
# (src/social_mpc_controller.cpp:172), window up to half the costmap's larger side (src/path_handler.cpp:69-71: 20 m / 2)
PLAN_WINDOW = (4.0, 10.0)


def sharded_ticks(ShardedEpisode, shards, ticks, B, prm, scenes, curv, device_index, plan, plan_len, tp):
    """config.closed_loop.sharded; a failure (e.g. a driver that refuses the graph capture) is reported, not raised."""
    import numpy as np

    try:
        se = ShardedEpisode(prm, scenes, curv, np.zeros((480, 480), np.uint32), np.array([-16.0, -16.0]),
                            float(np.float32(0.1)), device=device_index, plan=plan, plan_len=plan_len, traj_params=tp,
                            fov_angle=np.pi, shards=shards, graphs=True, plan_window=PLAN_WINDOW)
        for _ in range(2):
            se.tick()
        se.synchronize()
        t0 = time.perf_counter()
        for _ in range(ticks):
            se.tick()
        se.synchronize()
        shard_tick_s = (time.perf_counter() - t0) / ticks
        failures = int((se.gather("status") == 2).sum().item())
        del se
    except Exception as e:  # noqa: BLE001
        return {"shards": shards, "error": f"{type(e).__name__}: {e}"[:300]}
    return {"shards": shards, "hip_graphs": True, "ms_per_tick": shard_tick_s * 1e3, "ticks_per_s": B / shard_tick_s,
            "last_tick_failures": failures,
            "note": "the same robots as independent shards on separate streams (ShardedEpisode): one graph launch per shard "
                    "and tick, solve grids sized by smpc_set_solve_share, chain kernels at wave priority 3"}


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
                      fov_angle=np.pi,  # pi: everybody on the costmap is seen, the 8-agent workload of the headline config
                      plan_window=PLAN_WINDOW)
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
    # the same robots as three independent shards (streams), each tick of a shard one HIP graph launch
    from nav2_social_mpc_controller_amd.episode import ShardedEpisode
    shards = 3
    sharded = sharded_ticks(ShardedEpisode, shards, ticks, B, prm, scenes, curv, device_index, plan, plan_len, tp)
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
    stages["plan_window"] = {"kernel_ms": tm["window_ms"]}
    return {"ticks_per_s": B / tick_s, "ms_per_tick": tick_s * 1e3, "ticks_timed": ticks,
            "sharded": sharded,
            "chain": "transformGlobalPlan(f4) -> trajectorize(f3) -> fov filter + people_to_status(f4, f2) -> format_to_optimize(f2) -> project_people(f1) "
                     "-> solve(a1-a12, people block staged inside) -> memory store(f2)",
            "stages": stages, "last_tick_failures": int((ep.res["status"] == 2).sum().item()),
            "projection_errors": int((ep.proj_error != 0).sum().item())}


def shape_record(prm, B, N, device, local_rank, reps=3, parity_scenes=128, **scene_kw):
    """One BASELINE configuration measured like the headline one, single stream: the K1 sweep (staged people block,
    critic-major rows) and the solve launch (reference-layout input), HIP events on the launching stream."""
    import numpy as np
    import torch

    from nav2_social_mpc_controller_amd.scenes import make_scenes
    from nav2_social_mpc_controller_amd.solver import BatchSolver

    sc = make_scenes(prm, B, N, seed=0x5EED0001, **scene_kw)
    T = sc.T
    CH, bl, nb, P, M, _ = prm.dims(T, True)
    s = BatchSolver(prm, device=local_rank)
    sb, tens = sc.to_device(device)
    rb, rt = s.alloc_results(B, T, device)
    ms = []
    for _ in range(reps + 1):
        s.solve_device(sb, rb)
        ms.append(s.last_kernel_ms())
    solve_ms = float(min(ms[1:]))
    eo, et = s.alloc_eval(B, T, device, row_order=1)
    keep = s.stage_people_device(sb, device)
    stage_ms = s.last_kernel_ms()
    k1 = []
    for _ in range(reps + 2):
        s.eval_device(sb, tens["init_params"].data_ptr(), eo)
        k1.append(s.last_kernel_ms())
    k1_ms = float(np.median(k1[2:]))
    ev = rt["evaluations"].cpu().numpy().astype(np.int64)
    st = rt["status"].cpu().numpy()
    bytes_sweep = algorithmic_bytes_per_sweep(N, T, P, M)
    del keep
    # the oracle (checker) on the first scenes of this shape too: a sub-record that only said "failures: 0" proved nothing
    par = None
    if parity_scenes > 0:
        from oracle import oracle_py as O
        n = min(B, parity_scenes)
        rz = O.solve(prm, sc.select(np.arange(n)), nthreads=usable_cores(), theta_zero_convention=True)
        firm = rz["marginal_decisions"] == 0
        d = np.abs(rt["cmds"].cpu().numpy()[:n] - rz["cmds"]).reshape(n, -1).max(axis=1)
        par = {"scenes": int(n), "scenes_with_firm_decisions": int(firm.sum()),
               "max_abs_dcmd": float(d[firm].max()) if firm.any() else None, "scenes_over_1e-5": int((d[firm] > 1e-5).sum()),
               "iterations_equal_on_firm": int((rt["iterations"].cpu().numpy()[:n][firm] == rz["iterations"][firm]).sum())}
    return {"scenes": B, "people": N, "T": T, "P": P, "M": M, "slot_width": s.solve_slot_width(B, T, N),
            "k1_us": k1_ms * 1e3, "k1_frac_hbm": B * bytes_sweep / (k1_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "stage_people_us": stage_ms * 1e3,
            "solve_ms": solve_ms, "solves_per_s": B / (solve_ms * 1e-3), "mean_sweeps_per_solve": float(ev.mean()),
            "ns_per_scene_sweep": solve_ms * 1e6 / float(ev.sum()),
            "solve_frac_hbm_algorithmic": float(ev.sum()) * bytes_sweep / (solve_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "bytes_per_sweep": bytes_sweep, "failures": int((st == 2).sum()), "parity_sample": par}


def parity_sample(prm, scenes, out, n_sample, cores):
    """The oracle (CPU restatement, checker) on the first n_sample scenes of this rank's shard. `out`: the result tensors
    of the solve of those scenes."""
    import numpy as np

    from oracle import oracle_py as O
    sample = scenes.select(np.arange(n_sample))
    t1 = time.perf_counter()
    ref = O.solve(prm, sample, nthreads=cores)          # reference-literal oracle: also the timed CPU baseline
    cpu_s = time.perf_counter() - t1
    refz = O.solve(prm, sample, nthreads=cores, theta_zero_convention=True)   # checker (parity)
    got = out["cmds"].cpu().numpy()[:n_sample]
    got_cost = out["final_cost"].cpu().numpy()[:n_sample]
    dcmd = np.abs(got - refz["cmds"]).reshape(n_sample, -1).max(axis=1)
    firm = refz["marginal_decisions"] == 0
    clean = (ref["sign_noise_events"] == 0) & (ref["marginal_decisions"] == 0)
    dlit = np.abs(got - ref["cmds"]).reshape(n_sample, -1).max(axis=1)
    # every scene over the tolerance is listed with its cost gap: a scene that took another LM path must still end on an
    # equally good optimum (relative gap of the final costs; negative = the device's optimum is the lower one)
    over = np.where(dcmd > 1e-5)[0]
    gap = (got_cost - refz["final_cost"]) / np.maximum(refz["final_cost"], 1e-300)
    listed = [{"scene": int(i), "firm": bool(firm[i]), "abs_dcmd": float(dcmd[i]), "rel_cost_gap": float(gap[i]),
               "device_iterations": int(out["iterations"][i].item()), "oracle_iterations": int(refz["iterations"][i])}
              for i in over[np.argsort(-dcmd[over])][:16]]
    par = {"scenes": int(n_sample), "scenes_with_firm_decisions": int(firm.sum()),
           "max_abs_dcmd": float(dcmd[firm].max()), "median_abs_dcmd": float(np.median(dcmd)),
           "scenes_over_1e-5": int((dcmd[firm] > 1e-5).sum()),
           "scenes_over_1e-5_among_marginal": int((dcmd[~firm] > 1e-5).sum()),
           "scenes_over_1e-5_listed": listed,
           "worst_rel_cost_gap_over_1e-5": float(gap[over].max()) if len(over) else None,
           "checker": "CPU oracle (parity unpinned: restatement, not Ceres), theta:=0 convention for exactly equal velocities",
           "literal_oracle": {"scenes_without_sign_noise": int(clean.sum()),
                              "max_abs_dcmd_on_those": float(dlit[clean].max()) if clean.any() else None}}
    return par, cpu_s


def dry_run(args, rank, local_rank, world):
    """`--dry-run`: the launch / rendezvous / reduction plumbing of the N > 1 path on CPU ranks (gloo), with NO solve at
    all — not a measurement: value is null and the line says so. What it exercises is what an 8-GPU run adds to the
    1-GPU run: rank spawn, barrier, MAX / SUM reductions of the summary, the all_gather of the parameter blocks and
    rank 0's single JSON line. tests/test_bench_launch.py runs it with 2 ranks."""
    import numpy as np
    import torch
    import torch.distributed as dist

    from nav2_social_mpc_controller_amd import dist as D

    if os.environ.get("SMPC_BENCH_DRY_FAIL_RANK") == str(rank):
        sys.exit(3)   # failure injection for the launcher test: a rank that dies must fail the whole command
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)
    B, P = args.batch, 6
    lo, hi = D.weak_shard(B, rank)
    params = torch.arange(lo, hi, dtype=torch.float64).unsqueeze(1).repeat(1, P)   # stand-in: rows tagged with scene ids
    if dist.is_initialized():
        dist.barrier()
    t0 = time.perf_counter()
    time.sleep(0.01 * (rank + 1))   # ranks finish at different times: the summary must carry the MAX
    elapsed = time.perf_counter() - t0
    tmax = torch.tensor([elapsed], dtype=torch.float64)
    if dist.is_initialized():
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    summ = D.reduce_summary({"scenes": B, "max_rank": float(rank)})
    g = D.gather_params(params).numpy()
    line = None
    if rank == 0:
        line = {"metric": "MPC solves/sec (dry run of the launch path: nothing was solved)", "value": None, "unit": "solves/s",
                "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": None, "higher_is_better": True,
                "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "none", "dry_run": True,
                "config": {"workload": "dry run (gloo, CPU ranks): spawn + barrier + reductions + all_gather only"},
                "summary": summ, "elapsed_max_s": float(tmax.item()),
                "gather": {"ranks_seen": int(g.shape[0]), "params_shape": list(g.shape),
                           "first_scene_id_per_rank": [float(g[i, 0, 0]) for i in range(g.shape[0])],
                           "distinct_rank_slices": int(len({hash(g[i].tobytes()) for i in range(g.shape[0])}))}}
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()
    return line


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100,
                    help="timed steps (the four streams fill and drain inside the timed region: 40 steps 1.83 ms / step, 100: 1.79, 200: 1.77)")
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--batch", type=int, default=8192, help="scenes per GPU")
    ap.add_argument("--people", type=int, default=8)
    ap.add_argument("--fixed-iterations", type=int, default=0, help="1: run exactly 40 LM iterations per scene")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the extra measurements in `config` (PCIe-inclusive, fixed-40, other shapes, closed loop): profiling runs")
    ap.add_argument("--streams", type=int, default=4,
                    help="consecutive steps are issued round-robin on this many HIP streams (one solver handle and one "
                         "scene batch of its own each)")
    ap.add_argument("--dry-run", action="store_true",
                    help="exercise the N-rank launch / reduction path on CPU ranks (gloo) without solving anything")
    args = ap.parse_args()

    # `python bench.py --gpus N` launched plainly: become the launcher of N ranks (before any GPU call), relay rank 0's line
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))

    # Only the JSON line may reach stdout: libraries underneath (RCCL prints a version banner at communicator creation)
    # write to file descriptor 1 too. Keep the real stdout aside and point fd 1 at stderr for the rest of the run.
    sys.stdout.flush()
    real_stdout = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    from nav2_social_mpc_controller_amd import dist as D
    rank, local_rank, world = D.env_rank_world()
    if args.dry_run:
        line = dry_run(args, rank, local_rank, world)
        if line is not None:
            real_stdout.write(json.dumps(line) + "\n")
            real_stdout.flush()
        return

    import numpy as np
    import torch
    import torch.distributed as dist

    from nav2_social_mpc_controller_amd.params import OptimizerParams
    from nav2_social_mpc_controller_amd.scenes import make_scenes
    from nav2_social_mpc_controller_amd.solver import BatchSolver

    force_dist = os.environ.get("SMPC_BENCH_FORCE_DIST") == "1"   # rehearse the RCCL path with a single rank
    if args.gpus > 1 or world > 1 or force_dist:
        if world != args.gpus:
            raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
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
    n_streams = max(1, args.streams)
    # Consecutive steps (independent batches of a serving stream) are issued round-robin on a few HIP streams, one
    # solver handle each: the persistent solve kernel ends with a tail of a few long scenes (per-scene LM iteration
    # counts vary 5..40), and the next batch's waves fill the CUs that the tail leaves idle. Every stream solves a scene
    # batch of its OWN (scene ids disjoint across streams and ranks: 4 x 424 MB of inputs per GPU, more than the
    # 256 MiB Infinity Cache holds), so no step finds its costmaps or people blocks warm from the step before;
    # results of the steps of stream i land in result set i.
    lo, hi = D.weak_shard(B, rank)
    batches = [make_scenes(prm, B, N, seed=0x5EED0001, first_scene=lo + i * B * world) for i in range(n_streams)]
    scenes = batches[0]
    T = scenes.T
    CH, bl, nb, P, M, _ = prm.dims(T, True)

    solvers = [BatchSolver(prm, device=local_rank) for _ in range(n_streams)]
    from nav2_social_mpc_controller_amd.episode import concurrent_streams
    hip_streams = concurrent_streams(n_streams, device)  # streams that do not share a hardware queue (probed)
    for sv, st in zip(solvers, hip_streams):
        sv.set_stream(st.cuda_stream)
    solver = solvers[0]
    dev_batches = [sc.to_device(device) for sc in batches]
    sb, tens = dev_batches[0]
    results = [sv.alloc_results(B, T, device) for sv in solvers]
    rb, out = results[0]

    def barrier():
        torch.cuda.synchronize(device)
        if dist.is_initialized():
            dist.barrier()
        torch.cuda.synchronize(device)

    for w in range(max(args.warmup, n_streams)):
        solvers[w % n_streams].solve_device(dev_batches[w % n_streams][0], results[w % n_streams][0])
    barrier()
    # HIP events around every step, recorded on the stream that step is launched on (begin .. end = staging pass + solve
    # kernel of that step): per-launch durations and the span of the timed region as the device saw it
    ev0 = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    ev1 = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for k in range(args.steps):
        i = k % n_streams
        ev0[k].record(hip_streams[i])
        solvers[i].solve_device(dev_batches[i][0], results[i][0])
        ev1[k].record(hip_streams[i])
    barrier()
    elapsed = time.perf_counter() - t0
    step_ms = np.array([ev0[k].elapsed_time(ev1[k]) for k in range(args.steps)])
    span_ms = max(ev0[0].elapsed_time(e) for e in ev1)
    # per-launch device time of the solve kernel alone (library-side HIP events around its last launch per handle)
    solve_ms_each = [sv.last_kernel_ms() for sv in solvers[:min(n_streams, args.steps)]]
    solve_ms = float(sum(solve_ms_each) / len(solve_ms_each))
    # un-overlapped launches for reference (single stream, nothing else in flight): HIP events of the library around
    # the solve kernel alone, the average over launches is what a `rocprofv3 --stats` of a one-stream run reports
    solo = []
    for _ in range(5):
        solver.solve_device(sb, rb)
        torch.cuda.synchronize(device)
        solo.append(solver.last_kernel_ms())
    solo_ms = float(np.mean(solo[1:]))
    solo_min_ms = float(min(solo))

    tmax = torch.tensor([elapsed], dtype=torch.float64, device=device)
    if dist.is_initialized():
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    elapsed_max = float(tmax.item())

    evals_each = [r[1]["evaluations"].cpu().numpy().astype(np.int64) for r in results]
    evals = evals_each[0]
    iters = out["iterations"].cpu().numpy()
    status = out["status"].cpu().numpy()
    local = {"scenes": B, "sweeps": int(evals.sum()), "iterations": int(iters.sum()),
             "converged": int((status == 0).sum()), "no_convergence": int((status == 1).sum()),
             "failed": int((status == 2).sum()), "max_solve_kernel_ms": solve_ms, "max_solo_kernel_ms": solo_ms}
    summ = D.reduce_summary(local, device=device)

    # ---- correctness evidence on EVERY rank: the oracle on a sample of this rank's own shard, reduced with
    # MAX / SUM; the optimised parameters of all ranks gathered on rank 0
    cores = usable_cores()
    par = None
    cpu_s = None
    n_sample = 0
    if not args.no_cpu_baseline:
        share = max(1, cores // max(1, world))
        n_sample = min(B, 512 * share) if world == 1 else min(B, 512)
        par, cpu_s = parity_sample(prm, scenes, out, n_sample, share)
    gathered = None
    if dist.is_initialized():
        if par is not None:
            red = D.reduce_summary({"max_abs_dcmd": par["max_abs_dcmd"], "scenes": par["scenes"],
                                    "scenes_with_firm_decisions": par["scenes_with_firm_decisions"],
                                    "scenes_over_1e-5": par["scenes_over_1e-5"],
                                    "scenes_over_1e-5_among_marginal": par["scenes_over_1e-5_among_marginal"]}, device=device)
            par.update({k: (float(v) if k.startswith("max_") else int(v)) for k, v in red.items()})
            par["reduced_over_ranks"] = world
        gathered = D.gather_params(out["params"])      # [world][B][P] on every rank (all_gather over RCCL)

    # K1 stand-alone sweep for the roofline line: staged people block, critic-major rows (the coalesced store path)
    eo, eout = solver.alloc_eval(B, T, device, row_order=1)
    solver.eval_device(sb, tens["init_params"].data_ptr(), eo)        # reference-layout people: staging pass + K1
    k1_raw = []
    for _ in range(3):
        t1 = time.perf_counter()
        solver.eval_device(sb, tens["init_params"].data_ptr(), eo)
        torch.cuda.synchronize(device)
        k1_raw.append((time.perf_counter() - t1) * 1e3)
    keep = solver.stage_people_device(sb, device)
    stage_ms = solver.last_kernel_ms()
    k1 = []
    for _ in range(10):
        solver.eval_device(sb, tens["init_params"].data_ptr(), eo)
        k1.append(solver.last_kernel_ms())
    k1_ms = float(np.mean(k1[2:]))       # the average launch, as a rocprofv3 --stats summary reports it
    k1_min_ms = float(min(k1[2:]))
    eo.row_order = 0
    k1r = []
    for _ in range(5):
        solver.eval_device(sb, tens["init_params"].data_ptr(), eo)
        k1r.append(solver.last_kernel_ms())
    k1_ref_order_ms = float(np.mean(k1r[1:]))
    sb.people_records, sb.people_aux = None, None
    del keep

    if rank == 0:
        total_solves = summ["scenes"] * args.steps
        value = total_solves / elapsed_max
        bytes_sweep = algorithmic_bytes_per_sweep(N, T, P, M)
        sweeps_each = [int(e.sum()) for e in evals_each]
        sweeps_per_launch = float(np.mean(sweeps_each))        # the streams solve different scenes: mean over them
        eff_ms = elapsed_max / args.steps * 1e3          # what one launch costs the timed region
        measured_peak = solver.fp64_peak_tflops()
        # ---- the binding resource of the dominant kernel: FP64 VALU work / VALU instruction issue. Instruction counts per
        # launch come from the committed PMC passes (same scenes as stream 0 of rank 0), accepted only for the very
        # device sources running now and scaled by the sweeps this run's launches made; times are this run's.
        pmc, pmc_info = pmc_counters("smpc_solve_kernel")
        k1pmc, _ = pmc_counters("smpc_eval_kernel")
        fp64 = None
        top = {"bound": "fp64_valu", "kernel": "smpc_solve_kernel", "achieved": None, "peak": FP64_PEAK_TFS,
               "unit": "TFLOP/s", "frac": None, "traffic": None}
        if all(k in pmc for k in ("SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU")):
            scale = sweeps_per_launch / float(sweeps_each[0])   # PMC workload = stream 0's scenes
            flop = 64.0 * (pmc["SQ_INSTS_VALU_ADD_F64"] + pmc["SQ_INSTS_VALU_MUL_F64"] + 2.0 * pmc["SQ_INSTS_VALU_FMA_F64"]) * scale
            valu = pmc["SQ_INSTS_VALU"] * scale
            f64_instr = (pmc["SQ_INSTS_VALU_ADD_F64"] + pmc["SQ_INSTS_VALU_MUL_F64"] + pmc["SQ_INSTS_VALU_FMA_F64"]) * scale
            fp64 = {"flop_per_launch": flop, "valu_instructions_per_launch": valu,
                    "fp64_arithmetic_share_of_valu_instructions": f64_instr / valu,
                    "achieved_TFs_overlapped": flop / (eff_ms * 1e-3) / 1e12,
                    "achieved_TFs_lone_launch": flop / (solo_ms * 1e-3) / 1e12,
                    "frac_of_78.6_overlapped": flop / (eff_ms * 1e-3) / 1e12 / FP64_PEAK_TFS,
                    "frac_of_78.6_lone_launch": flop / (solo_ms * 1e-3) / 1e12 / FP64_PEAK_TFS,
                    "measured_peak_TFs": measured_peak,
                    # every wave64 VALU instruction (FP64 or not) holds its SIMD for 4 cycles: the issue-slot view
                    "valu_issue_frac_overlapped": valu * VALU_CYCLES / (SIMDS * 2.4e9 * eff_ms * 1e-3),
                    "valu_issue_frac_lone_launch": valu * VALU_CYCLES / (SIMDS * 2.4e9 * solo_ms * 1e-3),
                    "note": "counts: PMC passes committed in profiles/ for these very device sources (pmc.csrc_digest_*), scaled "
                            "by sweeps; times: this run (overlapped = timed region / launches, lone = mean of lone launches "
                            "by HIP events); issue fractions assume 2.4 GHz"}
            top.update({"achieved": fp64["achieved_TFs_overlapped"], "frac": fp64["frac_of_78.6_overlapped"],
                        "achieved_lone_launch": fp64["achieved_TFs_lone_launch"], "frac_lone_launch": fp64["frac_of_78.6_lone_launch"],
                        "valu_issue_frac": fp64["valu_issue_frac_overlapped"],
                        "valu_issue_frac_lone_launch": fp64["valu_issue_frac_lone_launch"]})
        if "FETCH_SIZE" in pmc and "WRITE_SIZE" in pmc:
            # HBM-side bytes per solve launch (KiB counters; the solve kernel's reads are 32-byte records and unaligned
            # dwords, not the 16 B / lane streaming pattern the guide's x2 FETCH_SIZE correction is calibrated on: reported
            # as counted)
            top["traffic"] = (pmc["FETCH_SIZE"] + pmc["WRITE_SIZE"]) * 1024.0
        k1_achieved = B * bytes_sweep / (k1_ms * 1e-3) / 1e9
        k1_traffic = None
        if "FETCH_SIZE" in k1pmc and "WRITE_SIZE" in k1pmc:
            # K1 streams its inputs 16 B / lane: FETCH_SIZE x 2 per the guide's gfx950 correction, WRITE_SIZE as counted
            k1_traffic = (2.0 * k1pmc["FETCH_SIZE"] + k1pmc["WRITE_SIZE"]) * 1024.0
        bytes_launch = sweeps_per_launch * bytes_sweep
        line = {
            "metric": "MPC solves/sec (horizon=18, 8 agents, max 40 LM iterations, Ceres termination rules)",
            "value": value, "unit": "solves/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": eff_ms, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"BASELINE configs[2] per GPU: batch={B} scenes/GPU, {N} people, horizon=18 "
                                   f"(T={T}, block=6, P={P}, M={M}), 200x200 u8 costmap per scene, DENSE_SCHUR, "
                                   f"max 40 LM iterations with Ceres termination rules"
                                   + (" DISABLED (fixed 40 iterations)" if args.fixed_iterations else "")
                                   + f"; consecutive steps overlapped on {n_streams} HIP streams, every stream on a "
                                     f"scene batch of its own",
                       "scenes_per_gpu": B, "people": N, "T": T, "P": P, "M": M, "streams": n_streams,
                       "distinct_scene_batches": n_streams,
                       "single_stream_solves_per_s_per_gpu": B / (summ["max_solo_kernel_ms"] * 1e-3),
                       "mean_lm_iterations": summ["iterations"] / summ["scenes"],
                       "mean_sweeps_per_solve": summ["sweeps"] / summ["scenes"],
                       "status": {"convergence": int(summ["converged"]), "no_convergence": int(summ["no_convergence"]),
                                  "failure": int(summ["failed"])}},
            "roofline": dict(top, **{
                "bound_note": "binding resource of the dominant kernel (the fused solve): FP64 VALU work under VALU instruction "
                              "issue; achieved = FP64 flop per launch / time per launch of THIS run (overlapped region; the "
                              "lone-launch figures beside it), peak = datasheet FP64 vector rate. The solve never writes J, so "
                              "its HBM traffic is a fraction of the contract's algorithmic bytes: the HBM figure of the "
                              "contract is K1's, under `contract`",
                "pmc": pmc_info,
                "launch_ms_effective": eff_ms, "launch_ms_lone_mean": solo_ms, "launch_ms_lone_min": solo_min_ms,
                "sweeps_per_launch": sweeps_per_launch, "sweeps_per_launch_by_stream": sweeps_each,
                "fp64_valu": fp64,
                "contract": {"bound": "hbm", "kernel": "smpc_eval_kernel (K1: the residual / Jacobian sweep, rows written)",
                             "achieved": k1_achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": k1_achieved / HBM_PEAK_GBS,
                             "traffic": k1_traffic,
                             "launch_ms_mean": k1_ms, "launch_ms_min": k1_min_ms,
                             "algorithmic_bytes_per_launch": B * bytes_sweep, "bytes_per_sweep": bytes_sweep,
                             "input": "staged people block (smpc_stage_people_batch), critic-major rows (row_order 1)",
                             "reference_row_order_launch_ms": k1_ref_order_ms,
                             "reference_row_order_frac": B * bytes_sweep / (k1_ref_order_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                             "stage_people_kernel_ms": stage_ms,
                             "wall_ms_from_reference_layout_people": float(min(k1_raw))},
                # the solve's sweeps priced as if each wrote its Jacobian like K1 does (it does not): kept for continuity
                # with earlier rounds, NOT a measured bandwidth
                "frac_algorithmic_if_J_were_written": bytes_launch / (eff_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "frac_algorithmic_if_J_were_written_lone_launch": bytes_launch / (solo_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "timed_region_hip_events": {
                    "steps": args.steps, "span_ms": float(span_ms), "mean_step_ms": float(step_ms.mean()),
                    "min_step_ms": float(step_ms.min()), "max_step_ms": float(step_ms.max()),
                    "steps_in_flight": float(step_ms.sum() / span_ms),
                    "solve_kernel_ms_last_launch_per_stream": solve_ms_each,
                    "note": "begin / end events of every step on its own stream: a step (staging pass + solve kernel) lasts "
                            "mean_step_ms while steps_in_flight of them overlap; launch_ms_effective = span / steps is what one "
                            "launch costs the region (rocprofv3 kernel trace of the same command: profiles/r03_bench_trace_*)"}}),
        }
        if par is not None:
            line["parity"] = par
        if gathered is not None:
            g = gathered.cpu().numpy()
            line["gather"] = {"ranks_seen": int(g.shape[0]), "params_shape": list(g.shape),
                              "all_finite": bool(np.isfinite(g).all()),
                              "rank0_slice_equals_local": bool(np.array_equal(g[0], out["params"].cpu().numpy())),
                              "distinct_rank_slices": int(len({hash(g[i].tobytes()) for i in range(g.shape[0])}))}
        if world == 1 and not args.no_cpu_baseline and par is not None:
            line["cpu_baseline"] = {"value": n_sample / cpu_s, "unit": "solves/s", "cores": max(1, cores), "kind": "port",
                                    "sample": f"first {n_sample} scenes of the same workload, one solve per thread "
                                              f"(CPU restatement oracle/smpc_oracle.cpp, not Ceres), {cpu_s:.1f} s"}
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
            del fixed
            # the other single-GPU BASELINE shapes, measured like the headline one but on one stream
            cfg5 = OptimizerParams.readme().replace(control_horizon=30, max_time=2.0)

            def extra(name, fn, *a, **kw):  # an extra that fails is reported, it does not cost the headline line
                try:
                    line["config"][name] = fn(*a, **kw)
                except Exception as e:  # noqa: BLE001
                    line["config"][name] = {"error": f"{type(e).__name__}: {e}"[:300]}

            extra("cfg2", shape_record, OptimizerParams.readme(), 1024, 4, device, local_rank)
            extra("cfg2_8192", shape_record, OptimizerParams.readme(), 8192, 4, device, local_rank)
            extra("cfg5", shape_record, cfg5, 8192, 16, device, local_rank)
            extra("params_yaml_n3", shape_record, OptimizerParams.params_yaml(), 8192, 3, device, local_rank)
            extra("soc_work_obst_benchmark_n3", shape_record, OptimizerParams.soc_work_obst_benchmark(), 8192, 3, device,
                  local_rank, map_cells=OptimizerParams.BENCHMARK_COSTMAP_CELLS)
            extra("closed_loop", closed_loop_extras, prm, scenes, local_rank)
        real_stdout.write(json.dumps(line) + "\n")
        real_stdout.flush()
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
