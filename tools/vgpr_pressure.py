"""Developer tool: rough VGPR liveness over the straight-line body of one kernel in a hipcc -S listing.
usage: vgpr_pressure.py file.s kernel_symbol_substring [step]   -> live VGPR count every `step` instructions (branches ignored)."""
import re
import sys

src, sym = sys.argv[1], sys.argv[2]
step = int(sys.argv[3]) if len(sys.argv) > 3 else 100
lines = open(src).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and sym in l and l.rstrip().endswith(":") or (l.startswith("_Z") and sym in l and ": " in l and ";" in l))
body = []
for l in lines[start + 1:]:
    if l.startswith("\t.section") or l.startswith(".Lfunc_end"):
        break
    t = l.split(";")[0].strip()
    if not t or t.endswith(":") or t.startswith("."):
        continue
    body.append(t)


def regs(tok):
    out = []
    for m in re.finditer(r"v\[(\d+):(\d+)\]|\bv(\d+)\b", tok):
        if m.group(1):
            out += list(range(int(m.group(1)), int(m.group(2)) + 1))
        else:
            out.append(int(m.group(3)))
    return out


defs, uses = [], []
for t in body:
    parts = t.split(None, 1)
    op = parts[0]
    ops = parts[1].split(",") if len(parts) > 1 else []
    d, u = [], []
    if op.startswith(("ds_write", "global_store", "scratch_store", "buffer_store", "s_", "ds_bpermute")) and not op.startswith("ds_bpermute"):
        for o in ops:
            u += regs(o)
    else:
        if ops:
            d += regs(ops[0])
        for o in ops[1:]:
            u += regs(o)
        if op.startswith(("v_fmac", "v_mac", "v_pk_fmac")) or "mfma" in op and len(ops) >= 4:
            pass
    defs.append(set(d))
    uses.append(set(u))
n = len(body)
live = set()
count = [0] * n
for i in range(n - 1, -1, -1):
    live -= defs[i]
    live |= uses[i]
    count[i] = len(live)
mx = max(range(n), key=lambda i: count[i])
print(f"{n} instructions, max live {count[mx]} at #{mx}: {body[mx][:80]}")
for i in range(0, n, step):
    j = max(range(i, min(n, i + step)), key=lambda k: count[k])
    print(f"{i:6d} {count[j]:4d}  {body[j][:70]}")
