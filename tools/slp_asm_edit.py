"""Developer tool used by tools/slp_variants.sh: named edits of the device assembly hipcc emits for gn_qkv.hip WITH the SLP vectoriser.
   python tools/slp_asm_edit.py in.s out.s <edit>"""
import re, sys
src, dst, name = sys.argv[1:4]
s = open(src).read()
SUSPECT = re.compile(r"^\tv_pk_fma_f32 v\[(\d+):(\d+)\], v\[(\d+):(\d+)\], v\[(\d+):(\d+)\], v\[(\d+):(\d+)\] op_sel:\[0,1,0\]$", re.M)
n = 0
if name == "base":
    pass
elif name == "scalar_fma":        # the suspect alone -> two v_fma_f32 (bit-identical arithmetic)
    s, n = SUSPECT.subn(r"\tv_fma_f32 v\1, v\3, v\6, v\7\n\tv_fma_f32 v\2, v\4, v\6, v\8", s)
elif name == "scalar_mul":        # the broadcast multiplies in front of it -> v_mul_f32
    s, n = re.subn(r"^\tv_pk_mul_f32 v\[(\d+):(\d+)\], v\[(\d+):(\d+)\], v\[(\d+):(\d+)\] op_sel_hi:\[0,1\]$", r"\tv_mul_f32 v\1, v\3, v\5\n\tv_mul_f32 v\2, v\3, v\6", s, flags=re.M)
elif name == "nop_before_pk":
    s, n = re.subn(r"^(\tv_pk_(mul|fma|add)_f32 )", r"\ts_nop 4\n\1", s, flags=re.M)
elif name == "nop_around_suspect":
    s, n = SUSPECT.subn(lambda m: "\ts_nop 7\n" + m.group(0) + "\n\ts_nop 7", s)
elif name == "opsel_copy":
    # keep the packed fma, drop the op_sel: the register that carried sin' into the preceding v_pk_mul_f32 (dead after it) receives cos', and the
    # fma reads it with op_sel_hi:[1,0,1] (both lanes <- low register), the form the FIRST rotary pair uses
    lines = s.split("\n")
    out = []
    for i, l in enumerate(lines):
        m = SUSPECT.match(l)
        if not m:
            out.append(l); continue
        d0, d1, a0, a1, b0, b1, c0, c1 = m.groups()
        w = None
        for k in range(i - 1, max(i - 8, 0), -1):
            mm = re.match(rf"^\tv_pk_mul_f32 v\[{c0}:{c1}\], v\[(\d+):(\d+)\], v\[{c0}:{c1}\] op_sel_hi:\[0,1\]$", lines[k])
            if mm:
                w = mm.groups(); break
        assert w, f"no producer for {l}"
        out.append(f"\tv_mov_b32_e32 v{w[0]}, v{b1}")
        out.append(f"\tv_pk_fma_f32 v[{d0}:{d1}], v[{a0}:{a1}], v[{w[0]}:{w[1]}], v[{c0}:{c1}] op_sel_hi:[1,0,1]")
        n += 1
    s = "\n".join(out)
elif name == "not_in_place":
    # the suspect writes its SOURCE 2 pair instead of its source 0 pair (then two moves): is destination == source 0 part of it?
    s, n = SUSPECT.subn(r"\tv_pk_fma_f32 v[\7:\8], v[\3:\4], v[\5:\6], v[\7:\8] op_sel:[0,1,0]\n\tv_mov_b32_e32 v\1, v\7\n\tv_mov_b32_e32 v\2, v\8", s)
else:
    raise SystemExit(f"unknown edit {name}")
open(dst, "w").write(s)
print(f"  {name}: {n} edits")
