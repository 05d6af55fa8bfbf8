"""Pins the oracle to the real Ceres Solver on a machine that has it (TEST INFRASTRUCTURE ONLY; SURVEY §8c escape hatch).

    make -C oracle ceres && python oracle/ceres_check.py [case ...]

For every committed golden case (tests/golden/<case>_scenes.npz) the scenes are written in the flat binary form
oracle/ceres_harness.cpp reads, the harness (the reference's functors under ceres::DynamicAutoDiffCostFunction +
ceres::Solve with the options of reference src/optimizer.cpp:117-131) solves them, and its optimum, iteration count and
termination are compared with what the restated oracle produced for the same scenes (tests/golden/<case>_expected.npz:
oracle_params / oracle_iterations / oracle_status, reference-literal semantics). The north-star tolerance applies:
max |x_ceres - x_oracle| <= 1e-5 per scene whose decisions were firm (oracle_marginal_decisions == 0, no sign noise).
Exit code 0 only if every case passes; the printed table is what DESIGN.md §2 should quote.

Cannot run in the image this repository is built in (no Ceres): the harness does not build there and this script says so."""
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HARNESS = os.path.join(HERE, "_build", "ceres_harness")


def dump(prm, sc, path):
    """The flat layout documented at the top of oracle/ceres_harness.cpp."""
    P = prm.dims(sc.T, True)[3]
    sc.validate(P)
    with open(path, "wb") as f:
        f.write(np.array([sc.B, sc.T, sc.N, sc.size_x, sc.size_y, 1 if sc.costmap_shared else 0], np.int32).tobytes())
        f.write(np.array([sc.dt, sc.resolution], np.float64).tobytes())
        f.write(bytes(prm.to_c()))
        for a in (sc.pose0, sc.init_params, sc.path_pts, sc.goal_yaw, sc.people, sc.has_people, sc.costmap, sc.costmap_origin):
            f.write(np.ascontiguousarray(a).tobytes())


def main():
    from conftest import GOLDEN_CASES, load_golden
    if not os.path.exists(HARNESS):
        print("oracle/_build/ceres_harness is not built (make -C oracle ceres; needs libceres-dev). Parity stays unpinned.")
        return 2
    cases = sys.argv[1:] or GOLDEN_CASES
    bad = 0
    print(f"{'case':32s} {'scenes':>6s} {'firm':>5s} {'max|dx| firm':>13s} {'iters equal':>11s} {'status equal':>12s}")
    for name in cases:
        prm, sc, exp = load_golden(name)
        with tempfile.TemporaryDirectory() as d:
            dump(prm, sc, os.path.join(d, "scenes.bin"))
            r = subprocess.run([HARNESS, os.path.join(d, "scenes.bin"), os.path.join(d, "out.txt")], capture_output=True, text=True)
            if r.returncode != 0:
                print(name, "harness failed:", r.stderr[-500:])
                bad += 1
                continue
            rows = np.loadtxt(os.path.join(d, "out.txt"), ndmin=2)
        status, iters, x = rows[:, 1].astype(int), rows[:, 2].astype(int), rows[:, 5:]
        firm = (exp["oracle_marginal_decisions"] == 0) & (exp["oracle_sign_noise_events"] == 0)
        dx = np.abs(x - exp["oracle_params"]).max(axis=1)
        ok_x = float(dx[firm].max()) if firm.any() else 0.0
        it_eq = int((iters[firm] == exp["oracle_iterations"][firm]).sum())
        st_eq = int((status[firm] == exp["oracle_status"][firm]).sum())
        print(f"{name:32s} {sc.B:6d} {int(firm.sum()):5d} {ok_x:13.3e} {it_eq:5d}/{int(firm.sum()):<5d} {st_eq:6d}/{int(firm.sum()):<5d}  {r.stderr.strip()}")
        if ok_x > 1e-5 or st_eq != firm.sum():
            bad += 1
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
