"""ORACLE -- test infrastructure only, never imported by the product path.

CPU (torch fp32) restatement of the matcher half of GISNav's PoseNode hot path:

  * PoseNode pre-processing             ros/gisnav/gisnav/core/pose_node.py:246-284
  * kornia 0.7.2 ``laf_from_center_scale_ori`` / ``get_laf_*`` [EXT, not vendored]
  * kornia 0.7.2 ``LightGlueMatcher.forward`` wrapper          [EXT]  (call site pose_node.py:285-287)
  * kornia 0.7.2 ``LightGlue(features="sift")`` forward        [EXT]  (config  pose_node.py:109-121)
  * match gather                        ros/gisnav/gisnav/core/pose_node.py:289-297

PARITY UNPINNED: kornia / the pretrained ``sift_lightglue.pth`` are absent from the
reference tree and from this container, and the reference's own tests pin no numeric
result of this path.  The restatement follows the published kornia /
cvg-LightGlue algorithm; its sub-functions are cross-checked in ``tests/`` against the
structurally identical code in the ``transformers`` package where the two coincide
(double-softmax, mutual filter, rotary, keypoint normalisation).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s cpu_baseline leg may
import this module.
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor

# kornia LightGlue defaults + features["sift"] + PoseNode params (pose_node.py:111-118)
N_LAYERS = 9
NUM_HEADS = 4
DESC_DIM = 256
INPUT_DIM = 128
HEAD_DIM = DESC_DIM // NUM_HEADS
FILTER_THRESHOLD = 0.5  # PoseNode.CONFIDENCE_THRESHOLD, pose_node.py:60


# --------------------------------------------------------------------------- kornia LAF helpers [EXT]
def laf_from_center_scale_ori(xy: Tensor, scale: Tensor, ori_deg: Tensor) -> Tensor:
    """kornia.feature.laf_from_center_scale_ori: LAF = [scale * R(ori) | xy].

    xy (B,N,2), scale (B,N,1,1), ori (B,N,1) in degrees -> (B,N,2,3).
    R = [[cos, sin], [-sin, cos]] (kornia ``angle_to_rotation_matrix``).
    Called at pose_node.py:267-276.
    """
    ang = ori_deg * (math.pi / 180.0)  # kornia deg2rad: x * pi / 180
    cos_a, sin_a = torch.cos(ang), torch.sin(ang)
    rot = torch.stack([cos_a, sin_a, -sin_a, cos_a], dim=-1).view(*ori_deg.shape, 2, 2).squeeze(2)
    return torch.cat([scale * rot, xy.unsqueeze(-1)], dim=3)


def get_laf_center(laf: Tensor) -> Tensor:
    return laf[..., 2]


def get_laf_scale(laf: Tensor) -> Tensor:
    """sqrt(|det(A)|) of the 2x2 part, shape (B,N,1,1)."""
    a = laf[..., 0:1, 0:1] * laf[..., 1:2, 1:2] - laf[..., 1:2, 0:1] * laf[..., 0:1, 1:2]
    return a.abs().sqrt()


def get_laf_orientation(laf: Tensor) -> Tensor:
    """degrees, shape (B,N,1): rad2deg(atan2(A01, A00))."""
    return (180.0 * torch.atan2(laf[..., 0, 1], laf[..., 0, 0]) / math.pi).unsqueeze(-1)


def rootsift(desc: Tensor) -> Tensor:
    """pose_node.py:278-284: sqrt(L1-normalise(desc)) per row (eps 1e-12 from F.normalize)."""
    return F.normalize(desc, dim=-1, p=1).sqrt()


# --------------------------------------------------------------------------- LightGlue pieces [EXT]
def normalize_keypoints(kpts: Tensor, size: Tensor) -> Tensor:
    """(kpts - size/2) / (max(size)/2); kpts (B,N,2), size (B,2) = (w,h)."""
    shift = size.float().to(kpts) / 2
    scale = size.max(1).values.float().to(kpts) / 2
    return (kpts - shift[:, None]) / scale[:, None, None]


def rotate_half(x: Tensor) -> Tensor:
    x = x.unflatten(-1, (-1, 2))
    x1, x2 = x.unbind(dim=-1)
    return torch.stack((-x2, x1), dim=-1).flatten(start_dim=-2)


def apply_cached_rotary_emb(freqs: Tensor, t: Tensor) -> Tensor:
    return (t * freqs[0]) + (rotate_half(t) * freqs[1])


def posenc(wr: Tensor, x: Tensor) -> Tensor:
    """LearnableFourierPositionalEncoding: (2,B,1,N,64) cached for all layers."""
    projected = F.linear(x, wr)
    cosines, sines = torch.cos(projected), torch.sin(projected)
    emb = torch.stack([cosines, sines], 0).unsqueeze(-3)
    return emb.repeat_interleave(2, dim=-1)


# Arithmetic class of the attention.  "fp32": what kornia's Attention computes on CPU (the default, and what every parity test compares
# with).  "half_sdpa": a CPU EMULATION of the branch PoseNode takes when `torch.cuda.is_available()` (pose_node.py:81, 108-121 put the matcher
# on "cuda"): kornia 0.7.2's Attention then casts q, k, v to half and calls F.scaled_dot_product_attention (flash kernel: q k^T and P v on
# fp16 operands with f32 accumulation, the probabilities rounded to fp16 before P v, the output rounded to fp16), and the cross block runs it
# twice (once per direction) instead of sharing one `sim`.  [EXT, restated; no CUDA device here to pin it on.]  Used only to give the fast
# GPU modes' mismatch counts their context: how many correspondence indices the REFERENCE'S OWN CPU -> CUDA move flips on the same inputs.
ATTENTION_MODE = "fp32"


class attention_mode:
    """with attention_mode("half_sdpa"): ...   (test infrastructure; module-wide switch, restored on exit)"""

    def __init__(self, mode: str):
        assert mode in ("fp32", "half_sdpa"), mode
        self.mode = mode

    def __enter__(self):
        global ATTENTION_MODE
        self.prev, ATTENTION_MODE = ATTENTION_MODE, self.mode
        return self

    def __exit__(self, *exc):
        global ATTENTION_MODE
        ATTENTION_MODE = self.prev


def _half(t: Tensor) -> Tensor:
    return t.to(torch.float16).to(torch.float32)


def _sdpa_half(q: Tensor, k: Tensor, v: Tensor, scale: float) -> Tensor:
    """Flash-style half SDPA emulated in f32: operands and probabilities rounded to fp16, f32 accumulation, fp16 output."""
    q, k, v = _half(q), _half(k), _half(v)
    sim = torch.einsum("...id,...jd->...ij", q, k) * scale
    m = sim.max(-1, keepdim=True).values
    p = torch.exp(sim - m)
    den = p.sum(-1, keepdim=True)
    out = torch.einsum("...ij,...jd->...id", _half(p), v) / den
    return _half(out)


def attention(q: Tensor, k: Tensor, v: Tensor) -> Tensor:
    """fp32 softmax(q k^T / sqrt(d)) v -- what kornia's Attention computes on CPU."""
    s = q.shape[-1] ** -0.5
    if ATTENTION_MODE == "half_sdpa":
        return _sdpa_half(q, k, v, s)
    sim = torch.einsum("...id,...jd->...ij", q, k) * s
    attn = F.softmax(sim, -1)
    return torch.einsum("...ij,...jd->...id", attn, v)


def _ffn(sd: Dict[str, Tensor], p: str, x: Tensor) -> Tensor:
    h = F.linear(x, sd[p + ".ffn.0.weight"], sd[p + ".ffn.0.bias"])
    h = F.layer_norm(h, (h.shape[-1],), sd[p + ".ffn.1.weight"], sd[p + ".ffn.1.bias"], 1e-5)
    h = F.gelu(h)  # exact erf GELU (nn.GELU default)
    return F.linear(h, sd[p + ".ffn.3.weight"], sd[p + ".ffn.3.bias"])


def self_block(sd: Dict[str, Tensor], i: int, x: Tensor, enc: Tensor) -> Tensor:
    p = f"transformers.{i}.self_attn"
    qkv = F.linear(x, sd[p + ".Wqkv.weight"], sd[p + ".Wqkv.bias"])
    qkv = qkv.unflatten(-1, (NUM_HEADS, -1, 3)).transpose(1, 2)
    q, k, v = qkv[..., 0], qkv[..., 1], qkv[..., 2]
    q = apply_cached_rotary_emb(enc, q)
    k = apply_cached_rotary_emb(enc, k)
    ctx = attention(q, k, v)
    msg = F.linear(ctx.transpose(1, 2).flatten(start_dim=-2), sd[p + ".out_proj.weight"], sd[p + ".out_proj.bias"])
    return x + _ffn(sd, p, torch.cat([x, msg], -1))


def cross_block(sd: Dict[str, Tensor], i: int, x0: Tensor, x1: Tensor) -> Tuple[Tensor, Tensor]:
    p = f"transformers.{i}.cross_attn"
    scale = HEAD_DIM ** -0.5
    lin = lambda name, t: F.linear(t, sd[f"{p}.{name}.weight"], sd[f"{p}.{name}.bias"])  # noqa: E731
    heads = lambda t: t.unflatten(-1, (NUM_HEADS, -1)).transpose(1, 2)  # noqa: E731
    qk0, qk1 = heads(lin("to_qk", x0)), heads(lin("to_qk", x1))
    v0, v1 = heads(lin("to_v", x0)), heads(lin("to_v", x1))
    if ATTENTION_MODE == "half_sdpa":   # kornia CrossBlock with flash: two SDPA calls on the UNSCALED qk (SDPA applies 1 / sqrt(d) itself)
        m0, m1 = _sdpa_half(qk0, qk1, v1, scale), _sdpa_half(qk1, qk0, v0, scale)
    else:
        qk0, qk1 = qk0 * scale ** 0.5, qk1 * scale ** 0.5
        sim = torch.einsum("bhid, bhjd -> bhij", qk0, qk1)
        attn01 = F.softmax(sim, dim=-1)
        attn10 = F.softmax(sim.transpose(-2, -1).contiguous(), dim=-1)
        m0 = torch.einsum("bhij, bhjd -> bhid", attn01, v1)
        m1 = torch.einsum("bhji, bhjd -> bhid", attn10.transpose(-2, -1), v0)
    m0, m1 = (t.transpose(1, 2).flatten(start_dim=-2) for t in (m0, m1))
    m0, m1 = lin("to_out", m0), lin("to_out", m1)
    x0 = x0 + _ffn(sd, p, torch.cat([x0, m0], -1))
    x1 = x1 + _ffn(sd, p, torch.cat([x1, m1], -1))
    return x0, x1


def sigmoid_log_double_softmax(sim: Tensor, z0: Tensor, z1: Tensor) -> Tensor:
    b, m, n = sim.shape
    certainties = F.logsigmoid(z0) + F.logsigmoid(z1).transpose(1, 2)
    scores0 = F.log_softmax(sim, 2)
    scores1 = F.log_softmax(sim.transpose(-1, -2).contiguous(), 2).transpose(-1, -2)
    scores = sim.new_full((b, m + 1, n + 1), 0)
    scores[:, :m, :n] = scores0 + scores1 + certainties
    scores[:, :-1, -1] = F.logsigmoid(-z0.squeeze(-1))
    scores[:, -1, :-1] = F.logsigmoid(-z1.squeeze(-1))
    return scores


def match_assignment(sd: Dict[str, Tensor], i: int, d0: Tensor, d1: Tensor) -> Tuple[Tensor, Tensor]:
    p = f"log_assignment.{i}"
    md0 = F.linear(d0, sd[p + ".final_proj.weight"], sd[p + ".final_proj.bias"])
    md1 = F.linear(d1, sd[p + ".final_proj.weight"], sd[p + ".final_proj.bias"])
    d = md0.shape[-1]
    md0, md1 = md0 / d ** 0.25, md1 / d ** 0.25
    sim = torch.einsum("bmd,bnd->bmn", md0, md1)
    z0 = F.linear(d0, sd[p + ".matchability.weight"], sd[p + ".matchability.bias"])
    z1 = F.linear(d1, sd[p + ".matchability.weight"], sd[p + ".matchability.bias"])
    return sigmoid_log_double_softmax(sim, z0, z1), sim


def filter_matches(scores: Tensor, th: float):
    max0, max1 = scores[:, :-1, :-1].max(2), scores[:, :-1, :-1].max(1)
    m0, m1 = max0.indices, max1.indices
    indices0 = torch.arange(m0.shape[1], device=m0.device)[None]
    indices1 = torch.arange(m1.shape[1], device=m1.device)[None]
    mutual0 = indices0 == m1.gather(1, m0)
    mutual1 = indices1 == m0.gather(1, m1)
    max0_exp = max0.values.exp()
    zero = max0_exp.new_tensor(0)
    mscores0 = torch.where(mutual0, max0_exp, zero)
    mscores1 = torch.where(mutual1, mscores0.gather(1, m1), zero)
    valid0 = mutual0 & (mscores0 > th)
    valid1 = mutual1 & valid0.gather(1, m1)
    m0 = torch.where(valid0, m0, -1)
    m1 = torch.where(valid1, m1, -1)
    return m0, m1, mscores0, mscores1


def canonical_state_dict(sd: Dict[str, Tensor]) -> Dict[str, Tensor]:
    """Accept both the checkpoint spelling (``self_attn.{i}.*``) and kornia's renamed
    ``transformers.{i}.self_attn.*``."""
    out = {}
    for k, v in sd.items():
        for kind in ("self_attn", "cross_attn"):
            if k.startswith(kind + "."):
                rest = k[len(kind) + 1:]
                idx, tail = rest.split(".", 1)
                k = f"transformers.{idx}.{kind}.{tail}"
                break
        out[k] = torch.as_tensor(v, dtype=torch.float32)
    return out


def lightglue_forward(
    sd: Dict[str, Tensor],
    kpts0: Tensor, kpts1: Tensor,
    desc0: Tensor, desc1: Tensor,
    scales0: Tensor, scales1: Tensor,
    oris0: Tensor, oris1: Tensor,
    size0: Tensor, size1: Tensor,
    n_layers: int = N_LAYERS,
    filter_threshold: float = FILTER_THRESHOLD,
    taps: Optional[dict] = None,
) -> Dict[str, Tensor]:
    """LightGlue._forward for features="sift", depth/width_confidence = -1 (no early exit,
    no pruning: pose_node.py:113-116).  All inputs carry a leading batch dim of 1."""
    k0 = normalize_keypoints(kpts0, size0).clone()
    k1 = normalize_keypoints(kpts1, size1).clone()
    k0 = torch.cat([k0, scales0.unsqueeze(-1), oris0.unsqueeze(-1)], -1)  # add_scale_ori
    k1 = torch.cat([k1, scales1.unsqueeze(-1), oris1.unsqueeze(-1)], -1)
    d0 = F.linear(desc0.contiguous(), sd["input_proj.weight"], sd["input_proj.bias"])
    d1 = F.linear(desc1.contiguous(), sd["input_proj.weight"], sd["input_proj.bias"])
    e0 = posenc(sd["posenc.Wr.weight"], k0)
    e1 = posenc(sd["posenc.Wr.weight"], k1)
    if taps is not None:
        taps["kpts0"], taps["kpts1"] = k0, k1
        taps["enc0"], taps["enc1"] = e0, e1
        taps["x_in0"], taps["x_in1"] = d0, d1
    for i in range(n_layers):
        d0 = self_block(sd, i, d0, e0)
        d1 = self_block(sd, i, d1, e1)
        if taps is not None:
            taps[f"self{i}_0"], taps[f"self{i}_1"] = d0, d1
        d0, d1 = cross_block(sd, i, d0, d1)
        if taps is not None:
            taps[f"layer{i}_0"], taps[f"layer{i}_1"] = d0, d1
    scores, sim = match_assignment(sd, n_layers - 1, d0, d1)
    m0, m1, ms0, ms1 = filter_matches(scores, filter_threshold)
    if taps is not None:
        taps["sim"], taps["scores"] = sim, scores
    return {"matches0": m0, "matches1": m1, "matching_scores0": ms0, "matching_scores1": ms1, "log_assignment": scores}


def lightglue_matcher_forward(
    sd: Dict[str, Tensor], desc1: Tensor, desc2: Tensor, lafs1: Tensor, lafs2: Tensor,
    hw1=None, hw2=None, n_layers: int = N_LAYERS, filter_threshold: float = FILTER_THRESHOLD,
    taps: Optional[dict] = None,
) -> Tuple[Tensor, Tensor]:
    """kornia LightGlueMatcher.forward (call: pose_node.py:285-287).

    desc (N,128) f32 RootSIFT, lafs (1,N,2,3).  Returns (scores (K,1) f32, idx (K,2) int64).
    NB image_size falls back to (max_x, max_y) of each side's keypoints when hw is None.
    """
    if desc1.shape[0] < 2 or desc2.shape[0] < 2:
        return desc1.new_zeros((0, 1)), torch.zeros((0, 2), dtype=torch.int64)
    kp1, kp2 = get_laf_center(lafs1), get_laf_center(lafs2)
    if desc1.dim() == 2:
        desc1 = desc1.unsqueeze(0)
    if desc2.dim() == 2:
        desc2 = desc2.unsqueeze(0)
    size1 = kp1.max(dim=1)[0].reshape(-1, 2) if hw1 is None else torch.tensor([[hw1[1], hw1[0]]], dtype=torch.float32)
    size2 = kp2.max(dim=1)[0].reshape(-1, 2) if hw2 is None else torch.tensor([[hw2[1], hw2[0]]], dtype=torch.float32)
    ori1 = get_laf_orientation(lafs1).reshape(1, -1) * math.pi / 180.0
    ori1 = torch.where(ori1 < 0, ori1 + 2.0 * math.pi, ori1)
    ori2 = get_laf_orientation(lafs2).reshape(1, -1) * math.pi / 180.0
    ori2 = torch.where(ori2 < 0, ori2 + 2.0 * math.pi, ori2)
    pred = lightglue_forward(
        sd, kp1, kp2, desc1, desc2,
        get_laf_scale(lafs1).reshape(1, -1), get_laf_scale(lafs2).reshape(1, -1),
        ori1, ori2, size1, size2, n_layers, filter_threshold, taps,
    )
    matches0, mscores0 = pred["matches0"], pred["matching_scores0"]
    valid = matches0 > -1
    matches = torch.stack([torch.where(valid)[1], matches0[valid]], -1)
    return mscores0[valid].reshape(-1, 1), matches


def pose_node_match(
    sd: Dict[str, Tensor],
    kp_q: Tensor, desc_q: Tensor, size_q: Tensor, angle_q: Tensor,
    kp_r: Tensor, desc_r: Tensor, size_r: Tensor, angle_r: Tensor,
    taps: Optional[dict] = None, filter_threshold: float = FILTER_THRESHOLD,
):
    """pose_node.py:246-297 with torch-CPU tensors: LAF build, RootSIFT, matcher, gather.

    Returns (mkp_q (K,2) f32, mkp_r (K,2) f32, scores (K,1), idx (K,2) int64).
    """
    with torch.inference_mode():
        laf_q = laf_from_center_scale_ori(kp_q.unsqueeze(0), size_q[None, :, None, None], angle_q[None, :, None])
        laf_r = laf_from_center_scale_ori(kp_r.unsqueeze(0), size_r[None, :, None, None], angle_r[None, :, None])
        dq, dr = rootsift(desc_q), rootsift(desc_r)
        scores, idx = lightglue_matcher_forward(sd, dq, dr, laf_q, laf_r, filter_threshold=filter_threshold, taps=taps)
        kq = get_laf_center(laf_q).squeeze(0)
        kr = get_laf_center(laf_r).squeeze(0)
        return kq[idx[:, 0]], kr[idx[:, 1]], scores, idx
