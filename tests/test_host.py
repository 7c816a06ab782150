"""CPU-side tests: the C ABI loads and exports every declared symbol, host logic, and the
world_size-2 gloo path of the pair-sharding helpers.  No GPU compute here."""
import ctypes
import os
import re
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    hdr = open(os.path.join(ROOT, "include", "gisnav_amd.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(gn_[a-z0-9_]+)\s*\(", hdr)))


def test_library_builds_loads_and_exports_every_header_symbol():
    from gisnav_amd import build
    build.build(verbose=False)
    from gisnav_amd import _lib
    lib = _lib.load()
    declared = _declared_symbols()
    assert len(declared) >= 18
    raw = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(raw, name), f"{name} declared in include/gisnav_amd.h but not exported"
    assert set(declared) == set(_lib.SIGNATURES), "python binding and header disagree"
    # the binary is bound to its sources: the digest of every csrc file, the public header and the compile flags is compiled into
    # the library, gn_version() shows it, and _lib.load() has just compared it with the tree's
    assert lib.gn_version().decode() == "gisnav_amd 0.3.0 gfx950 src:" + build.source_digest()
    assert _lib.library_digest() == build.source_digest() == lib.gn_source_digest().decode()


def test_a_library_built_from_other_sources_is_refused(tmp_path, monkeypatch):
    """A git-ignored libgisnav_amd.so travels with the tree to the GPU box: if it was built before the last source edit it would silently run
    yesterday's kernels.  `_lib.load` compares the digest compiled into the binary with the digest of the tree and refuses on a mismatch
    (GISNAV_AMD_ALLOW_STALE=1: developer A/B builds only).  Simulated by making the tree's digest differ, in a child interpreter."""
    import subprocess
    import sys
    from gisnav_amd import build
    build.build(verbose=False)
    code = ("import gisnav_amd.build as b; b.source_digest = lambda: '0' * 16\n"
            "from gisnav_amd import _lib\n"
            "try:\n    _lib.load(); print('LOADED')\nexcept _lib.GnError as e:\n    print('REFUSED', 'built from other sources' in str(e))\n")
    env = {k: v for k, v in os.environ.items() if k != "GISNAV_AMD_ALLOW_STALE"}
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True, env=env, timeout=300)
    assert "REFUSED True" in r.stdout, (r.stdout, r.stderr[-500:])
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True, env=dict(env, GISNAV_AMD_ALLOW_STALE="1"), timeout=300)
    assert "LOADED" in r.stdout, (r.stdout, r.stderr[-500:])


def test_shipped_device_code_has_no_low_lane_opsel_packed_fma(tmp_path):
    """`v_pk_fma_f32 ... op_sel:[.,1,.]` (the LOW lane multiplies by the HIGH register of source 1) is the instruction behind k_qkv's run-to-run
    differences when gn_qkv.hip is built with the SLP vectoriser (docs/DESIGN_HISTORY.md 12.5: on the MI355X the low lane intermittently returns source 2 alone in
    lanes 48..63; replacing that one instruction makes the kernel repeatable).  hipcc must not have emitted it anywhere in the library that ships."""
    import shutil
    import subprocess
    from gisnav_amd import build, _lib
    build.build(verbose=False)
    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    assert os.path.exists(objdump), "llvm-objdump is part of the ROCm image this library is built in: without it the shipped code cannot be checked"
    so = shutil.copy(os.path.join(ROOT, "gisnav_amd", "libgisnav_amd.so"), tmp_path / "lib.so")
    subprocess.run([objdump, "--offloading", os.path.basename(so)], cwd=tmp_path, check=True, capture_output=True)   # writes lib.so.N.hipv4-... beside it
    objs = sorted(f for f in os.listdir(tmp_path) if "hipv4" in f)
    assert len(objs) >= len(build.SOURCES) - 3, objs      # (a source without kernels has no bundle)
    packed = bad = 0
    for f in objs:
        dis = subprocess.run([objdump, "-d", f], cwd=tmp_path, check=True, capture_output=True, text=True).stdout
        for m in re.finditer(r"v_pk_fma_f32 (.*)", dis):
            packed += 1
            sel = re.search(r"op_sel:\[([01]),([01]),([01])\]", m.group(1))
            srcs = [o.strip() for o in re.split(r",\s*(?![^\[]*\])", m.group(1).split(" op_sel")[0])][1:]
            if sel and any(b == "1" and srcs[i].startswith("v") for i, b in enumerate(sel.groups())):
                bad += 1
    assert packed > 1000          # the check looked at real code
    assert bad == 0, f"{bad} v_pk_fma_f32 read a VGPR source through a low-lane op_sel bit"


def _disassemble_kernels(tmp_path, wanted):
    """{mangled kernel name: (metadata dict, [(opcode, [operands])])} for the kernels of the SHIPPED library whose mangled name contains one of `wanted`."""
    import shutil
    import subprocess
    from gisnav_amd import build
    build.build(verbose=False)
    llvm = "/opt/rocm/lib/llvm/bin"
    assert os.path.exists(f"{llvm}/llvm-objdump") and os.path.exists(f"{llvm}/llvm-readelf"), "the ROCm image's llvm tools are needed to check the shipped code"
    so = shutil.copy(os.path.join(ROOT, "gisnav_amd", "libgisnav_amd.so"), tmp_path / "lib.so")
    subprocess.run([f"{llvm}/llvm-objdump", "--offloading", os.path.basename(so)], cwd=tmp_path, check=True, capture_output=True)
    out = {}
    for f in sorted(f for f in os.listdir(tmp_path) if "hipv4" in f):
        notes = subprocess.run([f"{llvm}/llvm-readelf", "--notes", f], cwd=tmp_path, check=True, capture_output=True, text=True).stdout
        if not any(w in notes for w in wanted):
            continue
        meta = {}
        for blk in re.split(r"\n\s*- \.agpr_count:", notes)[1:]:
            kv = dict(re.findall(r"\.(agpr_count|name|private_segment_fixed_size|sgpr_spill_count|vgpr_count|vgpr_spill_count):\s*(\S+)", ".agpr_count:" + blk))
            if "name" in kv:
                meta[kv["name"]] = {k: (v if k == "name" else int(v)) for k, v in kv.items()}
        dis = subprocess.run([f"{llvm}/llvm-objdump", "-d", f], cwd=tmp_path, check=True, capture_output=True, text=True).stdout
        parts = re.split(r"\n[0-9a-f]+ <([^>]+)>:\n", dis)
        for i in range(1, len(parts), 2):
            name = parts[i]
            if not any(w in name for w in wanted) or name not in meta:
                continue
            ins = []
            for ln in parts[i + 1].split("\n"):
                ln = ln.split("//")[0].strip()
                m = re.match(r"^(\S+)\s*(.*)$", ln)
                if m:
                    ins.append((m.group(1), [o.strip() for o in re.split(r",\s*(?![^\[]*\])", m.group(2))] if m.group(2) else []))
            out[name] = (meta[name], ins)
    return out


def test_superpoint_convolutions_hold_no_packed_f32_with_a_scalar_operand(tmp_path):
    """`v_pk_fma_f32 D, V, s[n:n+1], V op_sel_hi:[1,0,1]` (the power-of-two scale of the split convolutions in an SGPR pair) returned
    garbage for one quad of lanes in k_sp_conv_s16's epilogue, about one tile in 10^4, only with 16 waves per CU: the epilogues are written on
    scalar fmas, and the library that ships must not contain a packed f32 instruction with a scalar source in these kernels -- whatever a later
    compiler (or an SLP pass) makes of the source."""
    ks = _disassemble_kernels(tmp_path, ["k_sp_conv_sILi", "k_sp_conv_s16ILb"])
    assert len(ks) >= 8, sorted(ks)
    for name, (meta, ins) in ks.items():
        assert meta["vgpr_spill_count"] == 0 and meta["private_segment_fixed_size"] == 0, (name, meta)
        assert sum(op.startswith("v_mfma") for op, _ in ins) >= 36, name
        bad = [(op, ops) for op, ops in ins if op.startswith(("v_pk_fma_f32", "v_pk_mul_f32", "v_pk_add_f32")) and any(re.match(r"^s(\[|\d)", o) for o in ops[1:])]
        assert not bad, (name, bad[:3])


def _reg_range(op):
    m = re.match(r"^([va])\[(\d+):(\d+)\]$", op) or re.match(r"^([va])(\d+)()$", op)
    if not m:
        return None
    lo = int(m.group(2))
    return m.group(1), lo, int(m.group(3)) if m.group(3) else lo


def test_shipped_asm_dependent_kernels_are_tamper_evident(tmp_path):
    """VERDICT r4 item 5.  `k_attn_pw` and `k_ffn128` are correct only as long as hipcc keeps allocating registers the way their hand-placed
    instruction streams assume (gn_attention_pw.hip: score MFMAs written in assembly with VGPR accumulators, dead accumulators kept allocated so that
    a late MFMA write-back cannot land on a recycled V^T fragment; gn_ffn128.hip: a straight line of ~17 k instructions that is 15 x slower when a
    register array falls into scratch memory).  Both are pinned by build flags only (build.py) -- so the disassembly of the library that SHIPS is
    checked here, on the CPU, before a GPU box ever sees a compiler upgrade:
      * k_attn_pw<F16, 0> (and the bf16 twin): no scratch instruction, no spill, 96 accumulation registers, exactly 128 MFMAs in the code;
        56 of them deliver the score tiles into VGPRs, and in the 16 instructions behind each of those no memory-return instruction
        (ds_read / buffer_load / global_load / scratch_load) writes a register of its destination block -- the hardware does not interlock
        an LDS return against an MFMA still in flight, which is the bug of DESIGN 10.2 (b) -- and no VALU instruction touches the block within
        the first 6 instructions (the hand-counted wait);
      * k_ffn128<0, composed, one tile per workgroup> (the bench's dominant kernel): 512 registers (256 + 256 accumulators), 2304 MFMAs, at most
        the two spilled VGPRs / five scratch instructions of round 4's build, none of them inside GEMM 1's stream of 1536 MFMAs."""
    ks = _disassemble_kernels(tmp_path, ["k_attn_pwILb1ELi0E", "k_attn_pwILb0ELi0E", "k_ffn128ILi0ELb1ELb0ELi", "k_ffn128ILi0ELb1ELb1ELi0E"])
    pw = [n for n in ks if "k_attn_pw" in n]
    assert len(pw) == 2, sorted(ks)
    for name in pw:
        meta, ins = ks[name]
        assert meta["agpr_count"] == 96 and meta["vgpr_spill_count"] == 0 and meta["private_segment_fixed_size"] == 0, meta
        assert not any(op.startswith("scratch_") for op, _ in ins), name
        mf = [j for j, (op, _) in enumerate(ins) if op.startswith("v_mfma")]
        assert len(mf) == 128, (name, len(mf))
        score = [(j, _reg_range(ins[j][1][0])) for j in mf if _reg_range(ins[j][1][0]) and _reg_range(ins[j][1][0])[0] == "v"]
        assert len(score) == 56 and all(hi - lo == 15 for _, (_, lo, hi) in score), (name, len(score))
        for j, (_, lo, hi) in score:
            for k in range(j + 1, min(j + 17, len(ins))):
                op, ops = ins[k]
                if op.startswith("v_mfma") or not ops:
                    continue
                d = _reg_range(ops[0])
                hit = d is not None and d[0] == "v" and not (d[2] < lo or d[1] > hi)
                if op.startswith(("ds_read", "ds_load", "buffer_load", "global_load", "scratch_load", "flat_load")):
                    assert not hit, f"{name}: {op} {ops[0]} lands in the destination v[{lo}:{hi}] of the score MFMA {k - j} instructions earlier"
                elif op.startswith("v_") and not op.startswith("v_cmp") and k - j <= 6:
                    touched = hit or any((r := _reg_range(o)) is not None and r[0] == "v" and not (r[2] < lo or r[1] > hi) for o in ops[1:])
                    assert not touched, f"{name}: {op} touches v[{lo}:{hi}] {k - j} instructions behind the score MFMA that writes it"
    # (tag, MFMAs in the code, spilled VGPRs at most, scratch instructions at most): the one-tile tail alone, with the self / cross projection fused behind it
    # (round 5: + 768 / 512 MFMAs), and the walking form
    # round 6: the last template parameter is the number of partial products (3: n1 = 1536 MFMAs in GEMM 1, 2304 in the tail; 2: 1024 / 1536)
    for tag, nmf, n1, ntail, spills, scratch in (("k_ffn128ILi0ELb1ELb0ELi0ELi3E", 2304, 1536, 2304, 2, 5), ("k_ffn128ILi0ELb1ELb0ELi1ELi3E", 3072, 1536, 2304, 3, 5),
                                                 ("k_ffn128ILi0ELb1ELb0ELi2ELi3E", 2816, 1536, 2304, 19, 12), ("k_ffn128ILi0ELb1ELb1ELi0ELi3E", 2304, 1536, 2304, 22, 21),
                                                 ("k_ffn128ILi0ELb1ELb0ELi0ELi2E", 1536, 1024, 1536, 4, 6), ("k_ffn128ILi0ELb1ELb0ELi1ELi2E", 2304, 1024, 1536, 4, 6),
                                                 ("k_ffn128ILi0ELb1ELb0ELi2ELi2E", 2048, 1024, 1536, 6, 8)):
        (name,) = [n for n in ks if tag in n]
        meta, ins = ks[name]
        assert meta["agpr_count"] == 256 and meta["vgpr_count"] == 512, meta
        assert meta["vgpr_spill_count"] <= spills and meta["sgpr_spill_count"] == 0, meta
        mf = [j for j, (op, _) in enumerate(ins) if op.startswith("v_mfma")]
        assert len(mf) == nmf, (name, len(mf))
        sc = [j for j, (op, _) in enumerate(ins) if op.startswith("scratch_")]
        assert len(sc) <= scratch, (name, len(sc))
        if "Lb0ELi" in tag:     # the one-tile forms the bench runs: nothing from scratch memory inside the first GEMM's weight stream, nor inside the projection's
            assert not any(mf[0] < j < mf[n1 - 1] for j in sc), (name, sc, mf[0], mf[n1 - 1])
            assert not any(j > mf[ntail] for j in sc if nmf > ntail), (name, sc, mf[ntail])
        if nmf > ntail:
            # round 5: the fused projection's epilogue is written component by component -- with the f32x4 expressions of k_qkv hipcc
            # emitted `v_pk_mul_f32 D, s[n:n+1], V op_sel_hi:[0,1]`, and the one-tile self form then stored garbage in (o.z, o.w) of lanes 12..15 /
            # 28..31 (deterministic; gone with scalar arithmetic).  No packed f32 arithmetic behind the projection's first MFMA (the tail's own row-wise epilogue, in front of it, keeps its v_pk_fma_f32).
            packed = [(j, op) for j, (op, _) in enumerate(ins) if j > mf[ntail] and op.startswith("v_pk_") and op.endswith("_f32")]
            assert not packed, (name, packed[:8])


def test_late_round5_kernels_hold_their_operands_in_registers(tmp_path):
    """DESIGN 12.8.  The kernels added late in round 5 win by keeping the NEXT tile in registers while the matrix pipe works on the current one:
    k_attn_f32_ks (its first build, on 64-key tiles, held that tile in scratch memory: 528 bytes per lane), the prefetching forms of LoFTR's
    convolutions at the rows-per-wave counts lf_conv launches for stride 1, and the 64-row exact-f32 GEMMs.  From the disassembly of the library
    that ships: no spill, no scratch segment, the MFMAs there; the two-workgroups-per-CU forms within 256 registers."""
    ks = _disassemble_kernels(tmp_path, ["k_attn_f32_ksILi4E", "k_gemm_f32_r64ILi", "k_gemm_f32_m64ILi", "k_lf_convILi3ELi1E", "k_lf_conv_hILi3ELi1E", "k_lf_conv1ILb0E"])
    assert sum("k_attn_f32_ks" in n for n in ks) == 1 and sum("k_gemm_f32_r64" in n for n in ks) >= 12 and sum("k_lf_conv" in n for n in ks) >= 12, sorted(ks)
    for name, (meta, ins) in ks.items():
        assert meta["vgpr_spill_count"] == 0 and meta["private_segment_fixed_size"] == 0, (name, meta)
        assert not any(op.startswith("scratch_") for op, _ in ins), name
        if "k_lf_conv1" not in name:
            assert sum(op.startswith("v_mfma") for op, _ in ins) >= 12, name
        if re.search(r"k_lf_conv(_h)?ILi3ELi1ELi[12]ELi32ELb1E", name):     # prefetching forms at one / two rows per wave: two workgroups per CU
            assert meta["vgpr_count"] <= 256, (name, meta["vgpr_count"])


def test_skinny_kernels_wait_for_their_lds_dma_rows_before_reading_them(tmp_path):
    """the small-grid kernels stage their token rows with `global_load_lds_dwordx4` written in assembly and wait for them with a hand-counted
    `s_waitcnt vmcnt(N)` = "everything but the N weight loads issued since" (gn_skinny.hip stage_wait).  That count is only right while the compiler leaves
    exactly those loads between the DMA and the wait: a load it sank below the wait would make the wait too weak and the first ds_read would see rows that
    have not landed.  Checked on the code that ships: walking every k_skinny_* kernel in program order, a staged group of DMA rows must have been retired
    by a wait whose N does not exceed the vector-memory instructions issued since the group's last row, before any LDS read follows."""
    ks = _disassemble_kernels(tmp_path, ["k_skinny_"])
    checked = 0
    for name, (_, ins) in ks.items():
        if not any(op.startswith("global_load_lds") for op, _ in ins):
            continue
        outstanding, since, groups = False, 0, 0
        for j, (op, args) in enumerate(ins):
            if op.startswith("global_load_lds"):
                if not outstanding:
                    groups += 1
                outstanding, since = True, 0
            elif op.startswith(("global_load", "buffer_load", "flat_load", "scratch_load", "global_store", "buffer_store", "flat_store", "scratch_store", "global_atomic", "buffer_atomic")):
                since += 1
            elif op == "s_waitcnt":
                m = re.search(r"vmcnt\((\d+)\)", " ".join(args))
                if m and outstanding and int(m.group(1)) <= since:
                    outstanding = False
            elif op.startswith(("ds_read", "ds_load")):
                assert not outstanding, (name, j, "an LDS read follows LDS-DMA rows that no s_waitcnt has retired", since)
            elif op in ("s_endpgm", "s_branch", "s_setpc_b64"):
                outstanding, since = False, 0
        assert groups >= 1 and not outstanding, (name, groups)
        checked += 1
    assert checked >= 6, sorted(ks)


def test_product_path_has_no_cpu_fallback():
    from gisnav_amd import _lib
    from gisnav_amd.engine import PoseEngine
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(_lib.GnError):
        PoseEngine(0)
    from gisnav_amd.matcher import LightGlueMatcher
    with pytest.raises(_lib.GnError):
        LightGlueMatcher("sift", params={"depth_confidence": -1, "width_confidence": -1}).to("cpu")
    from gisnav_amd.loftr import LoFTR
    with pytest.raises(_lib.GnError):
        LoFTR(state_dict={"x": torch.zeros(1)}).to("cpu")                      # no CPU path
    with pytest.raises(_lib.GnError):
        LoFTR(pretrained="outdoor").to("cuda:0")                               # no checkpoint offline: fails loudly, never random weights


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "gisnav_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in src and "from oracle" not in src, f
    bench = open(os.path.join(ROOT, "bench.py")).read()
    # bench.py may touch the oracle only inside cpu_baseline()
    head, tail = bench.split("def cpu_baseline", 1)
    body, rest = tail.split("\ndef main", 1)
    assert "oracle" not in head and "oracle" not in rest.replace("oracle/", "")


def test_matcher_constructor_mirrors_reference_arguments():
    from gisnav_amd.matcher import LightGlueMatcher
    m = LightGlueMatcher("sift", params={"n_layers": 9, "filter_threshold": 0.5, "depth_confidence": -1, "width_confidence": -1})
    assert m.params["n_layers"] == 9 and m.params["filter_threshold"] == 0.5
    with pytest.raises(NotImplementedError):
        LightGlueMatcher("superpoint")
    with pytest.raises(NotImplementedError):   # adaptive depth/width is dead code in the reference
        LightGlueMatcher("sift", params={"depth_confidence": 0.99, "width_confidence": 0.99})


def test_weight_layout_matches_kornia_state_dict():
    from gisnav_amd.weights import canonical_state_dict, expected_shapes, synthetic_state_dict
    sd = synthetic_state_dict(0)
    shp = expected_shapes()
    for k, s in shp.items():
        assert sd[k].shape == s and sd[k].dtype == np.float32, k
    n_params = sum(int(np.prod(s)) for s in shp.values())
    assert abs(n_params - 11.88e6) / 11.88e6 < 0.06            # 11.88 M parameters
    w = sd["input_proj.weight"]
    assert np.allclose(w.T @ w, np.eye(128), atol=1e-5)
    c = canonical_state_dict({"self_attn.0.Wqkv.weight": sd["transformers.0.self_attn.Wqkv.weight"]})
    assert list(c) == ["transformers.0.self_attn.Wqkv.weight"]


def test_synthetic_pairs_are_deterministic_and_well_formed():
    from gisnav_amd.synthetic import IMG_H, IMG_W, K_MATRIX, make_pair
    a, b = make_pair(5), make_pair(5)
    assert np.array_equal(a.desc_q, b.desc_q) and np.array_equal(a.kp_r, b.kp_r) and np.array_equal(a.dem, b.dem)
    assert a.desc_q.shape == (1024, 128) and a.desc_q.dtype == np.float32
    assert (a.desc_q == np.rint(a.desc_q)).all() and a.desc_q.min() >= 0 and a.desc_q.max() <= 255
    assert a.dem.shape == (IMG_H, IMG_W) and a.dem.dtype == np.uint8 and a.dem.max() <= 40
    assert (a.kp_q[:, 0] >= 0).all() and (a.kp_q[:, 0] < IMG_W + 2).all()
    assert K_MATRIX[0, 0] == 205.4696 and K_MATRIX[1, 2] == 240.0
    m = a.gt_q2r >= 0
    assert m.sum() >= 300
    # the true matches re-project through the ground-truth pose to within the 0.5 px noise
    x, y = np.floor(a.kp_r[a.gt_q2r[m]]).astype(int).T
    obj = np.column_stack([a.kp_r[a.gt_q2r[m]], a.dem[y, x]]).astype(np.float64)
    cam = obj @ a.R_gt.T + a.t_gt.T
    uv = cam[:, :2] / cam[:, 2:] * K_MATRIX[0, 0] + K_MATRIX[:2, 2]
    assert np.abs(uv - a.kp_q[m]).max() < 3.0


def test_shard_range_partitions_contiguously():
    from gisnav_amd.dist import shard_range
    for total, world in [(256, 8), (32, 1), (64, 2), (10, 4), (3, 8)]:
        parts = [shard_range(total, r, world) for r in range(world)]
        flat = [i for p in parts for i in p]
        assert flat == list(range(total))
        assert max(len(p) for p in parts) == -(-total // world)
    assert shard_range(256, 3, 8) == range(96, 128)             # 32 pairs per GPU, BASELINE configs[3]


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    from gisnav_amd import dist as gd
    r, lr, w = gd.init("gloo")
    dev = torch.device("cpu")
    sd = {"a.weight": np.full((3, 2), float(rank + 1), np.float32), "b.bias": np.arange(4, dtype=np.float32) * (rank + 1)}
    sd = gd.broadcast_state_dict(sd, dev, src=0)
    shard = gd.shard_range(8, r, w)
    B = len(shard)
    out = dict(R=torch.eye(3, dtype=torch.float64).repeat(B, 1, 1) * (r + 1), t=torch.full((B, 3, 1), float(r), dtype=torch.float64),
               ok=torch.ones(B, dtype=torch.uint8), n_match=torch.full((B,), 100 + r, dtype=torch.int32),
               n_inliers=torch.full((B,), 90 + r, dtype=torch.int32))
    rec = gd.gather_records(gd.pack_records(shard.start, out))
    gd.barrier()
    mx = gd.max_over_ranks(float(r) + 0.5, dev)
    sm = gd.sum_over_ranks(float(B), dev)
    q.put((r, sd["a.weight"].tolist(), sd["b.bias"].tolist(), rec.numpy().tolist(), mx, sm))
    torch.distributed.destroy_process_group()


def test_two_process_gloo_broadcast_gather_and_timing_reduction():
    world, port = 2, 29611
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for r, a, b, rec, mx, sm in res:
        assert a == [[1.0, 1.0]] * 3 and b == [0.0, 1.0, 2.0, 3.0]          # rank 0's weights everywhere
        rec = np.array(rec)
        assert rec.shape == (8, 16)
        assert rec[:, 0].tolist() == list(range(8))                           # pair index: contiguous shards
        assert rec[:4, 2].tolist() == [100.0] * 4 and rec[4:, 2].tolist() == [101.0] * 4
        assert rec[5, 4] == 2.0 and rec[5, 13] == 1.0
        assert mx == 1.5 and sm == 8.0


def _worker8(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    from gisnav_amd import dist as gd
    r, lr, w = gd.init("gloo")
    dev = torch.device("cpu")
    shard = gd.shard_range(256, r, w)                       # BASELINE configs[3]: 256 pairs over 8 ranks
    B = len(shard)
    idx = torch.arange(shard.start, shard.stop, dtype=torch.float64)
    out = dict(R=torch.eye(3, dtype=torch.float64).repeat(B, 1, 1) * idx[:, None, None], t=idx[:, None, None].repeat(1, 3, 1),
               ok=torch.ones(B, dtype=torch.uint8), n_match=(idx % 7).to(torch.int32), n_inliers=(idx % 5).to(torch.int32))
    rec = gd.gather_records(gd.pack_records(shard.start, out))
    gd.barrier()
    q.put((r, B, shard.start, rec[:, 0].tolist(), rec[:, 13].tolist(), gd.max_over_ranks(0.1 * (r + 1), dev), gd.sum_over_ranks(float(B), dev)))
    torch.distributed.destroy_process_group()


def test_eight_process_gloo_shards_of_256_pairs():
    """World size 8 (the BASELINE node): contiguous 32-pair shards, one gather of the 256 result records, max / sum reductions."""
    world, port = 8, 29633
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker8, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=240) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for r, B, start, pair_idx, t0, mx, sm in res:
        assert B == 32 and start == 32 * r
        assert pair_idx == [float(i) for i in range(256)] and t0 == [float(i) for i in range(256)]
        assert abs(mx - 0.8) < 1e-12 and sm == 256.0


def test_bench_json_contract_fields_present_in_source():
    src = open(os.path.join(ROOT, "bench.py")).read()
    for key in ('"metric"', '"value"', '"unit"', '"n_gpus"', '"steps"', '"warmup"', '"ms_per_step"', '"higher_is_better"',
                '"scaling"', '"vs_baseline"', '"dtype"', '"data"', '"config"', '"roofline"', '"cpu_baseline"', '"workload"',
                '"traffic"', '"extra_configs"', '"inputs_resident"', '"debug_variant"', '"pcie_inclusive"', '"multi_gpu"', '"end_to_end"'):
        assert key in src, key


def test_bench_refuses_more_gpus_than_visible_and_mismatched_world():
    """`python bench.py --gpus N` must never silently run fewer ranks than asked for.  Without a launcher it spawns the
    ranks itself -- and refuses (exit code 2, no JSON line) when the box shows fewer GPUs; under a launcher whose WORLD_SIZE differs from
    --gpus it refuses as well.  (On this GPU-less container both paths stop before any device work.)"""
    import subprocess
    if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
        pytest.skip("box has >= 2 GPUs: the same command would run the 2-rank bench")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 2 and "refusing" in r.stderr and not any(l.startswith("{") for l in r.stdout.splitlines())
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"], env=dict(env, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0"),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 2 and "does not match --gpus 4" in r.stderr


def _worker_ident(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    from gisnav_amd import dist as gd
    gd.init("gloo")
    q.put((rank, gd.gather_strings(f"{rank}:gpu{rank}")))
    torch.distributed.destroy_process_group()


def test_two_process_gloo_rank_identity_gather():
    """bench.py's liveness check: every rank contributes `rank:device-identity`; the line is only printed when N ranks on N devices answered."""
    from gisnav_amd import dist as gd
    world, port = 2, gd.free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_ident, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert all(ids == ["0:gpu0", "1:gpu1"] for _, ids in res)


def test_cpulist_parser_of_the_staging_threads_numa_pinning():
    from gisnav_amd.engine import parse_cpulist
    assert parse_cpulist("0-3,8,10-11\n") == {0, 1, 2, 3, 8, 10, 11}
    assert parse_cpulist("5") == {5} and parse_cpulist("") is None
