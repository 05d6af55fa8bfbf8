"""Compact view of a rocprofv3 kernel-trace CSV: every smpc_* dispatch with begin / end relative to the first one, its
queue, and the number of solve kernels in flight when it started; plus the span / overlap summary of the timed region.
usage: python tools/trace_extract.py <..._kernel_trace.csv>"""
import csv
import sys

rows = []
with open(sys.argv[1]) as fh:
    for r in csv.DictReader(fh):
        name = r.get("Kernel_Name", "")
        if "smpc" not in name:
            continue
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name.split("(")[0].replace("void smpc::", ""),
                     r.get("Queue_Id", "?")))
rows.sort()
t0 = rows[0][0]
solve = [(a, b) for a, b, n, q in rows if "solve_kernel" in n]
print("# begin_us end_us dur_us queue kernel solve_kernels_in_flight_at_begin")
for a, b, n, q in rows:
    inflight = sum(1 for (x, y) in solve if x <= a < y)
    print(f"{(a - t0) / 1e3:10.1f} {(b - t0) / 1e3:10.1f} {(b - a) / 1e3:9.1f} {q:>4} {n} {inflight}")
if solve:
    # the timed region = the last `steps` solve launches; summary over all traced solve launches after the warm-up gap
    durs = [(b - a) / 1e3 for a, b in solve]
    span = (max(b for a, b in solve) - min(a for a, b in solve)) / 1e3
    print(f"# solve launches {len(solve)}: mean duration {sum(durs) / len(durs):.1f} us, span first-begin..last-end {span:.1f} us, "
          f"span / launches {span / len(solve):.1f} us, mean launches in flight {sum(durs) / span:.2f}")
