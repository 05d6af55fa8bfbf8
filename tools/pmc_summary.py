"""Collapse rocprofv3 --pmc counter_collection CSVs into one per-kernel table (mean per dispatch)."""
import csv, glob, os, sys, collections
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
    for k, cs in acc.items():
        fo.write(f"== {k}\n")
        for c in sorted(cs):
            v = cs[c]
            fo.write(f"  {c:28s} mean/dispatch {sum(v)/len(v):.6g}  (n={len(v)})\n")
print(open(os.path.join(out, "pmc_summary.txt")).read())
