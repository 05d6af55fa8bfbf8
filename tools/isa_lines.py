"""Per-basic-block instruction mix of a kernel compiled to assembly with -gline-tables-only, with the source lines each
block comes from. Development aid for the instruction diet of the solve kernel.
usage: python tools/isa_lines.py file.s <kernel symbol substring> [min_valu]"""
import collections
import re
import sys

path, sym = sys.argv[1], sys.argv[2]
min_v = int(sys.argv[3]) if len(sys.argv) > 3 else 8
files = {}
blocks = []  # (label, [(op, args, file, line)])
cur = None
loc = (0, 0)
inside = False
for l in open(path):
    s = l.strip()
    m = re.match(r"\.file\s+(\d+)\s+\"([^\"]*)\"(?:\s+\"([^\"]*)\")?", s)
    if m:
        files[int(m.group(1))] = (m.group(3) or m.group(2)).split("/")[-1]
        continue
    if re.match(r"^[A-Za-z_.$][\w.$]*:", s):
        lab = s.split(":")[0]
        if sym in lab and not lab.startswith("."):
            inside = True
        elif inside and not lab.startswith(".L"):
            inside = False
        if inside and not lab.startswith(".Ltmp"):  # .Ltmp labels only carry line-table entries
            cur = [lab, []]
            blocks.append(cur)
        continue
    if not inside:
        continue
    if s.startswith(".loc"):
        p = s.split()
        loc = (int(p[1]), int(p[2]))
        continue
    if s.startswith(".") or s.startswith(";") or not s:
        continue
    op = s.split()[0]
    cur[1].append((op, s, loc[0], loc[1]))
    if op.startswith("s_cbranch") or op in ("s_branch", "s_endpgm", "s_setpc_b64"):
        cur = [cur[0] + "+", []]
        blocks.append(cur)

F64 = ("v_fma_f64", "v_mul_f64", "v_add_f64", "v_fmac_f64")
tot = collections.Counter()
for lab, ins in blocks:
    v = [i for i in ins if i[0].startswith("v_")]
    if len(v) < min_v:
        continue
    f = sum(1 for i in v if i[0].startswith(F64))
    sal = sum(1 for i in ins if i[0].startswith("s_"))
    ds = sum(1 for i in ins if i[0].startswith("ds_"))
    gl = sum(1 for i in ins if i[0].startswith(("global_", "buffer_", "scratch_", "flat_")))
    lines = collections.Counter((files.get(i[2], str(i[2])), i[3]) for i in v)
    byfile = collections.defaultdict(list)
    for (fn, ln), c in lines.items():
        byfile[fn].append(ln)
    where = " ".join(f"{fn.replace('smpc_', '').replace('.hpp', '')}:{min(ls)}-{max(ls)}" for fn, ls in byfile.items())
    other = collections.Counter(i[0] for i in v if not i[0].startswith(F64))
    print(f"{lab[:14]:14s} V{len(v):4d} f64 {f:4d} S{sal:3d} D{ds:3d} G{gl:2d} | {where}")
    print("      other:", " ".join(f"{k.replace('v_', '')}:{c}" for k, c in other.most_common(10)))
