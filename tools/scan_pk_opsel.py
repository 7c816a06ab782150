"""Developer tool: which packed-f32 instructions with a LOW-lane op_sel bit does hipcc emit for each source file (its own build flags)?
`v_pk_fma_f32 ... op_sel:[0,1,0]` is the instruction behind k_qkv's run-to-run differences when gn_qkv.hip is built with the SLP vectoriser
(docs/DESIGN_HISTORY.md 12.5): this lists every occurrence in the library.   python tools/scan_pk_opsel.py [--slp-qkv]"""
import collections, os, re, subprocess, sys, tempfile
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from gisnav_amd import build as B
pat = re.compile(r"^\s*(v_pk_(?:fma|mul|add)_f32)\s+(.*?)\s+op_sel:\[([01,]+)\]", re.M)
total = 0
for src in B.SOURCES:
    flags = list(B.FLAGS) + list(B.EXTRA_FLAGS.get(src, []))
    if "--slp-qkv" in sys.argv and src == "gn_qkv.hip":
        flags = [f for f in flags if f != "-fno-slp-vectorize"]
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "k.s")
        subprocess.run(["/opt/rocm/bin/hipcc", *flags, "--cuda-device-only", "-S", os.path.join(B.CSRC, src), "-o", out], check=True, stderr=subprocess.DEVNULL)
        s = open(out).read()
    c = collections.Counter()
    for m in pat.finditer(s):
        ops = [o.strip() for o in re.split(r",\s*(?![^\[]*\])", m.group(2))]
        sel = m.group(3).split(",")
        for i, bit in enumerate(sel):            # source i of the instruction is ops[i + 1]
            if bit == "1":
                c[f"{m.group(1)} src{i} {'vgpr' if ops[i + 1].startswith('v') else 'sgpr' if ops[i + 1].startswith('s') else 'other'}"] += 1
    npk = len(re.findall(r"^\s*v_pk_(?:fma|mul|add)_f32", s, re.M))
    print(f"{src:22s} packed f32 instructions {npk:5d};  with a low-lane op_sel bit: {dict(c) if c else 'none'}")
    total += sum(v for k, v in c.items() if k.endswith("vgpr"))
print("VGPR sources read through a low-lane op_sel bit:", total)
