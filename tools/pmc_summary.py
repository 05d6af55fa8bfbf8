"""Collapse rocprofv3 --pmc counter_collection CSVs into one per-kernel table (mean per dispatch)."""
import csv, glob, os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nav2_social_mpc_controller_amd.buildinfo import csrc_digest
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(out, "*", "**", "*counter_collection.csv"), recursive=True):
    with open(f) as fh:
        per_dispatch = collections.defaultdict(float)
        names = {}
        for row in csv.DictReader(fh):
            key = (row["Dispatch_Id"], row["Counter_Name"])
            per_dispatch[key] += float(row["Counter_Value"])
            names[row["Dispatch_Id"]] = row["Kernel_Name"]
        for (did, cname), v in per_dispatch.items():
            k = names[did]
            if "smpc" in k:
                acc[k.split("(")[0]][cname].append(v)
with open(os.path.join(out, "pmc_summary.txt"), "w") as fo:
    # identity of the device sources these counters belong to (bench.py refuses them for any other build)
    fo.write(f"# csrc_digest: {csrc_digest()}\n")
    fo.write(f"# workload: {os.environ.get('SMPC_PMC_WORKLOAD', 'tools/prof_target.py (cfg3: 8192 scenes, 8 people; 3 K1 sweeps + 2 solves)')}\n")
    for k, cs in acc.items():
        fo.write(f"== {k}\n")
        for c in sorted(cs):
            v = cs[c]
            fo.write(f"  {c:28s} mean/dispatch {sum(v)/len(v):.6g}  (n={len(v)})\n")
print(open(os.path.join(out, "pmc_summary.txt")).read())
