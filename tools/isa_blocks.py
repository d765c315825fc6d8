"""Dev tool: basic blocks of one kernel in a hipcc -S listing with VALU / SALU / LDS / VMEM counts, and the backward
branches (loops) among them.  python tools/isa_blocks.py frr.s '<mangled kernel name substring>'"""
import re, sys
path, key = sys.argv[1], sys.argv[2]
lines = open(path).read().splitlines()
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and key in l and ":" in l)
end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
blocks, cur = [], {"name": "entry", "line": start, "valu": 0, "salu": 0, "lds": 0, "vmem": 0, "br": []}
order = {}
for i in range(start + 1, end + 1):
    l = lines[i].strip()
    m = re.match(r"^(\.LBB[0-9_]+):", l)
    if m:
        blocks.append(cur); cur = {"name": m.group(1), "line": i, "valu": 0, "salu": 0, "lds": 0, "vmem": 0, "br": []}
        continue
    if not l or l.startswith(";") or l.startswith("."): continue
    op = l.split()[0]
    if op.startswith("v_"): cur["valu"] += 1
    elif op.startswith("ds_"): cur["lds"] += 1
    elif op.startswith(("global_", "buffer_", "flat_", "scratch_")): cur["vmem"] += 1
    elif op.startswith("s_"):
        cur["salu"] += 1
        if op.startswith(("s_cbranch", "s_branch")): cur["br"].append(l.split()[1])
blocks.append(cur)
idx = {b["name"]: k for k, b in enumerate(blocks)}
tot = [sum(b[k] for b in blocks) for k in ("valu", "salu", "lds", "vmem")]
print("kernel lines %d..%d  blocks %d  VALU %d SALU %d LDS %d VMEM %d" % (start, end, len(blocks), *tot))
loops = []
for k, b in enumerate(blocks):
    for t in b["br"]:
        if t in idx and idx[t] <= k: loops.append((idx[t], k))
loops.sort(key=lambda x: (x[0], -x[1]))
for a, z in loops:
    s = [sum(b[k] for b in blocks[a:z + 1]) for k in ("valu", "salu", "lds", "vmem")]
    print("loop %-12s .. %-12s lines %5d-%5d  VALU %4d SALU %4d LDS %3d VMEM %3d" % (blocks[a]["name"], blocks[z]["name"], blocks[a]["line"] - start, blocks[z]["line"] - start, *s))
if len(sys.argv) > 3:
    for k, b in enumerate(blocks):
        print("%4d %-12s line %5d VALU %4d SALU %3d LDS %3d VMEM %3d -> %s" % (k, b["name"], b["line"] - start, b["valu"], b["salu"], b["lds"], b["vmem"], ",".join(b["br"])))
