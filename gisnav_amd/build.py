"""Build recipe for libgisnav_amd.so (gfx950 only, in-tree so the .so travels to the GPU box).

    python -m gisnav_amd.build [--force]

hipcc cross-compiles without a GPU.  The product path has no fallback: if this library is
missing, importing `gisnav_amd._lib` raises.
"""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libgisnav_amd.so")
SOURCES = ["gn_api.hip", "gn_gemm.hip", "gn_gemm_p2.hip", "gn_ffn.hip", "gn_ffn128.hip", "gn_qkv.hip", "gn_skinny.hip", "gn_attention.hip", "gn_attention_pw.hip", "gn_prep.hip", "gn_match_head.hip", "gn_knn.hip", "gn_warp.hip", "gn_geo.hip", "gn_sift.hip", "gn_superpoint.hip", "gn_pnp.hip", "gn_loftr.hip"]
# SLP vectoriser: ON for every file except gn_qkv.hip.  With it, hipcc (ROCm 7.2) packs the second rotary pair of k_qkv's epilogue
# (o.z = v.z cos' - v.w sin', o.w = v.w cos' + v.z sin') into `v_pk_fma_f32 D, A, B, C op_sel:[0,1,0]` -- the LOW lane multiplies by the HIGH
# register of B -- and on the MI355X that instruction, in this kernel, intermittently returns C alone in the low lane (the product is dropped)
# in lanes 48..63: o.z of ~0.1 % of the (token, pair) positions, different ones every run (tools/slp_diag.py).  Pinned by editing the SLP
# build's assembly one thing at a time (tools/slp_variants.sh, 20 bench-sized runs each): the suspect alone replaced by two v_fma_f32 -> 0 of
# 20 runs differ; kept packed but reading cos' from a low register (op_sel_hi:[1,0,1], the first pair's form) -> 0 of 20; s_nop 7 on both
# sides of it, destination != source 0, every wait forced to 0, the neighbouring v_pk_mul_f32 scalarised -> still 20 of 20.  A standalone
# loop of the same instruction (tools/probes/pk_fma_opsel.hip, 1.7e10 results next to MFMA waves) never fails, so it is legal code that
# misbehaves in this kernel's company, not a missing wait state or a barrier protocol.  Without the vectoriser the compiler never emits a
# low-lane op_sel on a packed fma; tests/test_host.py disassembles the shipped library and fails if one appears anywhere.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=on", "-Wno-unused-value"]
# attention: keep the MFMA accumulators in VGPRs (the softmax reads the scores and rescales the output every tile;
# with the default AGPR form each tile paid ~255 v_accvgpr_read/write moves on the VALU, the kernel's bottleneck)
EXTRA_FLAGS = {"gn_attention.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"],
               "gn_qkv.hip": ["-fno-slp-vectorize"],
               "gn_skinny.hip": ["-fno-slp-vectorize"],     # k_qkv's epilogue expressions (same bits): same flag
               # k_ffn128 places its VALU work instruction by instruction beside the MFMAs: packed f32 arithmetic is an anti-lever there
               # (and is one straight line of ~17 k instructions: the default size limit of `#pragma unroll` would silently leave a loop rolled, and its
               # register arrays in scratch memory)
               "gn_ffn128.hip": ["-fno-slp-vectorize", "-mllvm", "-pragma-unroll-threshold=1000000"],
               # k_attn_pw: the same style (one wave per SIMD, pinned order); accumulators in AGPRs (the default MFMA form)
               "gn_attention_pw.hip": ["-fno-slp-vectorize", "-mllvm", "-pragma-unroll-threshold=1000000"],
               # SIFT: no fused multiply-adds -- every float operation rounds separately, as in the oracle (and in OpenCV's scalar code)
               "gn_sift.hip": ["-ffp-contract=off"]}


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def source_digest() -> str:
    """16 hex digits over everything that decides the device code: every csrc/*.hip and *.h, the public header, and the compile flags.  The -m gpu
    parity tests stamp their report with it and bench.py prints the report's mismatch counts only as belonging to the build whose digest matches."""
    import hashlib
    h = hashlib.sha256()
    files = sorted(f for f in os.listdir(CSRC) if f.endswith((".hip", ".h")))
    for f in files + [os.path.join("..", "..", "include", "gisnav_amd.h")]:
        with open(os.path.join(CSRC, f), "rb") as fh:
            h.update(os.path.basename(f).encode() + b"\0" + fh.read() + b"\0")
    h.update(repr((FLAGS, sorted(EXTRA_FLAGS.items()))).encode())
    return h.hexdigest()[:16]


BUILD_ID_SRC = os.path.join(CSRC, "gn_build_id.hip")
BUILD_ID_OBJ = os.path.join(CSRC, "gn_build_id.o")
BUILD_ID_STAMP = os.path.join(CSRC, "gn_build_id.stamp")


def _build_id(digest: str, verbose: bool) -> bool:
    """Compile the digest into the library (gn_source_digest).  Returns True when the object was (re)built."""
    have = open(BUILD_ID_STAMP).read().strip() if os.path.exists(BUILD_ID_STAMP) else ""
    if have == digest and os.path.exists(BUILD_ID_OBJ):
        return False
    cmd = [_hipcc(), *FLAGS, f'-DGN_SOURCE_DIGEST="{digest}"', "-c", BUILD_ID_SRC, "-o", BUILD_ID_OBJ]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    with open(BUILD_ID_STAMP, "w") as f:
        f.write(digest + "\n")
    return True


def build(force: bool = False, verbose: bool = True) -> str:
    hdrs = [os.path.join(CSRC, "gn_common.h"), os.path.join(HERE, "..", "include", "gisnav_amd.h")]
    objs = []
    procs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(CSRC, src.replace(".hip", ".o"))
        objs.append(o)
        if force or _stale(o, [s] + hdrs):
            cmd = [_hipcc(), *FLAGS, *EXTRA_FLAGS.get(src, []), "-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd), flush=True)
            procs.append((src, subprocess.Popen(cmd)))
    failed = [src for src, p in procs if p.wait() != 0]
    if failed:
        raise RuntimeError(f"hipcc failed for {failed}")
    new_id = _build_id(source_digest(), verbose)
    objs.append(BUILD_ID_OBJ)
    if force or procs or new_id or _stale(LIB, objs):
        cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
