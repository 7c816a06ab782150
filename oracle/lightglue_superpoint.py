"""ORACLE -- test infrastructure only, never imported by the product path.

CPU (torch fp32) restatement of kornia 0.7.2 ``LightGlue(features="superpoint")`` -- the matcher of BASELINE.json configs[4]
("SuperPoint+LightGlue 1024-keypoint path"), which the reference tree itself does not contain: it differs
from the ``"sift"`` variant PoseNode instantiates (pose_node.py:109-121, restated in ``oracle/lightglue_sift.py``) only in
``input_dim = 256 = descriptor_dim`` (``input_proj`` is the identity) and ``add_scale_ori = False`` (the learnable Fourier
positional encoding sees the normalised (x, y) only, ``posenc.Wr`` is [32, 2]).  Blocks, match head and filter are the same code.

PINNED against importable third-party code: ``tests/test_oracle_pins.py`` loads the same weights into ``transformers``'
``LightGlueForKeypointMatching`` (a SuperPoint-LightGlue port; its separate cross-attention q / k projections are both set to
the shared ``to_qk``) and requires identical matches and scores from ``_match_image_pair``.
"""
from __future__ import annotations

from typing import Dict, Optional

import torch

from . import lightglue_sift as lg

Tensor = torch.Tensor


def lightglue_forward(sd: Dict[str, Tensor], kpts0: Tensor, kpts1: Tensor, desc0: Tensor, desc1: Tensor, size0: Tensor, size1: Tensor,
                      n_layers: int = lg.N_LAYERS, filter_threshold: float = 0.1, taps: Optional[dict] = None) -> Dict[str, Tensor]:
    """kpts (1,N,2) pixel coordinates, desc (1,N,256), size (1,2) = (w, h) of each image."""
    k0 = lg.normalize_keypoints(kpts0, size0).clone()
    k1 = lg.normalize_keypoints(kpts1, size1).clone()
    d0, d1 = desc0.contiguous(), desc1.contiguous()          # input_proj = Identity
    e0 = lg.posenc(sd["posenc.Wr.weight"], k0)
    e1 = lg.posenc(sd["posenc.Wr.weight"], k1)
    if taps is not None:
        taps["enc0"], taps["enc1"] = e0, e1
    for i in range(n_layers):
        d0 = lg.self_block(sd, i, d0, e0)
        d1 = lg.self_block(sd, i, d1, e1)
        d0, d1 = lg.cross_block(sd, i, d0, d1)
        if taps is not None:
            taps[f"layer{i}_0"], taps[f"layer{i}_1"] = d0, d1
    scores, sim = lg.match_assignment(sd, n_layers - 1, d0, d1)
    m0, m1, ms0, ms1 = lg.filter_matches(scores, filter_threshold)
    if taps is not None:
        taps["sim"], taps["scores"] = sim, scores
    return {"matches0": m0, "matches1": m1, "matching_scores0": ms0, "matching_scores1": ms1, "log_assignment": scores}


def match(sd: Dict[str, Tensor], kpts0: Tensor, desc0: Tensor, kpts1: Tensor, desc1: Tensor, hw0=None, hw1=None,
          filter_threshold: float = 0.1, taps: Optional[dict] = None):
    """kornia ``LightGlueMatcher("superpoint").forward`` packing: (scores (K,1), idx (K,2) int64).  kpts (N,2), desc (N,256);
    hw = (h, w) of the image or None -> keypoint extent (kornia's fallback)."""
    if desc0.shape[0] < 2 or desc1.shape[0] < 2:
        return desc0.new_zeros((0, 1)), torch.zeros((0, 2), dtype=torch.int64)
    k0, k1 = kpts0[None], kpts1[None]
    s0 = k0.max(dim=1)[0].reshape(-1, 2) if hw0 is None else torch.tensor([[hw0[1], hw0[0]]], dtype=torch.float32)
    s1 = k1.max(dim=1)[0].reshape(-1, 2) if hw1 is None else torch.tensor([[hw1[1], hw1[0]]], dtype=torch.float32)
    with torch.inference_mode():
        pred = lightglue_forward(sd, k0, k1, desc0[None], desc1[None], s0, s1, filter_threshold=filter_threshold, taps=taps)
    m0, ms0 = pred["matches0"], pred["matching_scores0"]
    valid = m0 > -1
    return ms0[valid].reshape(-1, 1), torch.stack([torch.where(valid)[1], m0[valid]], -1)
