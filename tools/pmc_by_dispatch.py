"""Print the counters of every dispatch whose kernel name contains argv[2] (default "solve_kernel") of a rocprofv3
--pmc run, in dispatch order."""
import csv, glob, collections, sys
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
acc = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    if (sys.argv[2] if len(sys.argv) > 2 else "solve_kernel") in r["Kernel_Name"]:
        acc.setdefault(int(r["Dispatch_Id"]), collections.defaultdict(float))[r["Counter_Name"]] += float(r["Counter_Value"])
for d in sorted(acc):
    print(d, {k: f"{v:.4g}" for k, v in acc[d].items()})
