"""Ad-hoc GPU parity probe (development tool): HIP path vs CPU oracle on seeded scenes."""
import sys, time, json
import numpy as np
sys.path.insert(0, ".")
from nav2_social_mpc_controller_amd.params import OptimizerParams
from nav2_social_mpc_controller_amd.scenes import make_scenes
from nav2_social_mpc_controller_amd.solver import BatchSolver
from oracle import oracle_py as O

def run(name, p, B, N, **kw):
    sc = make_scenes(p, B, N, **kw)
    CH, bl, nb, P, M, _ = p.dims(sc.T, True)
    s = BatchSolver(p)
    # K1 parity at the initial point and a perturbed point
    rng = np.random.default_rng(1)
    for tag, x in (("init", sc.init_params), ("pert", sc.init_params + 0.05 * rng.standard_normal(sc.init_params.shape))):
        eo = O.evaluate(p, sc, x)
        eg = s.evaluate(sc, x)
        dr = np.abs(eo["residuals"] - eg["residuals"]); dj = np.abs(eo["jacobian"] - eg["jacobian"])
        sr = np.maximum(1.0, np.abs(eo["residuals"])); sj = np.maximum(1.0, np.abs(eo["jacobian"]))
        print(f"[{name}] eval {tag}: max rel |dr| {np.max(dr/sr):.3e}  max rel |dJ| {np.max(dj/sj):.3e}  cost rel {np.max(np.abs(eo['cost']-eg['cost'])/np.maximum(1,eo['cost'])):.3e}")
        if np.max(dj/sj) > 1e-6:
            b, m, q = np.unravel_index(np.argmax(dj/sj), dj.shape)
            print("   worst J at scene", b, "row", m, "col", q, eo["jacobian"][b, m], eg["jacobian"][b, m], "r", eo["residuals"][b,m], eg["residuals"][b,m])
    t = time.time(); ro = O.solve(p, sc, nthreads=8, theta_zero_convention=True); to = time.time() - t
    t = time.time(); rg = s.solve(sc); tg = time.time() - t
    dc = np.abs(ro["cmds"] - rg["cmds"]).reshape(B, -1).max(axis=1)
    print(f"[{name}] solve: oracle {to:.2f}s gpu(host-staged) {tg:.3f}s kernel {s.last_kernel_ms():.3f} ms")
    print(f"[{name}] status oracle {np.bincount(ro['status'], minlength=3)} gpu {np.bincount(rg['status'], minlength=3)}")
    print(f"[{name}] iters oracle mean {ro['iterations'].mean():.2f} gpu mean {rg['iterations'].mean():.2f}; evals gpu mean {rg['evaluations'].mean():.1f}")
    print(f"[{name}] max|dcmd| {dc.max():.3e}; scenes >1e-5: {(dc>1e-5).sum()}/{B}; median {np.median(dc):.3e}; iter mismatch {(ro['iterations']!=rg['iterations']).sum()}")
    print(f"[{name}] max|dpath| {np.abs(ro['path']-rg['path']).max():.3e} final cost rel diff max {np.max(np.abs(ro['final_cost']-rg['final_cost'])/np.maximum(1,ro['final_cost'])):.3e}")
    return dc

if __name__ == "__main__":
    p = OptimizerParams.readme()
    run("cfg2-N4", p, 256, 4)
    run("cfg3-N8", p, 256, 8)
    run("ref-N3", p, 128, 3, n_valid=2)
    run("nopeople", p, 64, 3, people_present=False)
    p5 = p.replace(control_horizon=30, max_time=2.0)
    run("cfg5-N16", p5, 64, 16)
    py = OptimizerParams.params_yaml()
    run("params.yaml-N3", py, 64, 3)
