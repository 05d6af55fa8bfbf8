"""Parameter generators shared by tools/gpu_soak.py and tools/gpu_soak_case.py (development probes)."""
import sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from test_gpu_parity import _random_params as random_params  # noqa: E402,F401


def wide_params(rng):
    base = random_params(rng)
    bl = int(rng.integers(2, 6))
    nb = int(rng.integers(5, 11))
    mt = float(rng.choice([1.5, 2.0, 2.5, 3.0]))
    T = int(round(mt / 0.05)) - 2
    ch = min(nb * bl, T)  # control horizon inside the rollout
    return base.replace(control_horizon=ch, parameter_block_length=bl, max_time=mt)
