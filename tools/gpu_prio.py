"""Development probe: lone launch and 4-stream rate of the headline batch under SMPC_PRIO_STEP values (attained-service
wave priority of the solve kernel)."""
import os, subprocess, sys
vals = sys.argv[1:] or ["0", "16", "24", "32", "48"]
for v in vals:
    env = dict(os.environ, SMPC_PRIO_STEP=v, PROBED_STREAMS="1")
    r = subprocess.run([sys.executable, "tools/gpu_solveab.py", ""], env=env, capture_output=True, text=True)
    print("SMPC_PRIO_STEP=%s: %s" % (v, r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-400:]), flush=True)
