"""Basic-block summary of one kernel of the built library: instruction counts per block (VALU / SALU / LDS / VMEM) and
the block each branch goes to. Development aid for counting the instructions of a hot loop.
usage: python tools/isa_blocks.py <mangled kernel symbol> [min_valu]"""
import re, subprocess, sys, tempfile, os, glob
sym = sys.argv[1]
min_v = int(sys.argv[2]) if len(sys.argv) > 2 else 0
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = os.path.join(root, "nav2_social_mpc_controller_amd", "csrc", "libsmpc_hip.so")
objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
with tempfile.TemporaryDirectory() as d:
    subprocess.run(["cp", lib, d + "/lib.so"], check=True)
    subprocess.run([objdump, "--offloading", "lib.so"], cwd=d, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    co = glob.glob(d + "/lib.so*gfx950")[0]
    text = subprocess.run([objdump, "-d", co, "--disassemble-symbols=" + sym], check=True, capture_output=True, text=True).stdout
ins = []
for l in text.splitlines():
    m = re.match(r"\s+(\S+)\s*(.*?)\s*//\s*([0-9A-F]+):", l)
    if m:
        ins.append((int(m.group(3), 16), m.group(1), m.group(2)))
idx = {a: i for i, (a, _, _) in enumerate(ins)}
tgt = {}
for i, (a, op, args) in enumerate(ins):
    if op.startswith("s_cbranch") or op == "s_branch":
        off = int(args.split()[0])
        off -= 65536 if off >= 32768 else 0
        tgt[i] = idx.get(a + 4 + 4 * off)
leaders = sorted(l for l in ({0} | {t for t in tgt.values() if t is not None} | {i + 1 for i in tgt}) if l < len(ins))
print(f"{len(ins)} instructions, {len(leaders)} blocks")
for b, st in enumerate(leaders):
    en = leaders[b + 1] if b + 1 < len(leaders) else len(ins)
    ops = [ins[i][1] for i in range(st, en)]
    v = sum(o.startswith("v_") for o in ops)
    if v < min_v:
        continue
    s = sum(o.startswith("s_") for o in ops)
    d_ = sum(o.startswith("ds_") for o in ops)
    g = sum(o.startswith(("global_", "buffer_", "flat_", "scratch_")) for o in ops)
    print(f"B{st:5d}-{en:5d} V{v:4d} S{s:4d} D{d_:3d} G{g:3d}  {ins[en - 1][1]} -> {tgt.get(en - 1)}")
