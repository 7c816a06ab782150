"""ORACLE -- test infrastructure only, never imported by the product path.

CPU (torch fp32) restatement of the SuperPoint extractor named by BASELINE.json configs[4] ("SuperPoint+LightGlue 1024-keypoint
path") -- the "conv backbone" of north_star.  The reference tree does not contain it; the restatement
follows DeTone et al.'s architecture as published and as ported in ``transformers`` (modeling_superpoint.py), and is PINNED to
that port: ``tests/test_superpoint.py`` loads the same weights into ``SuperPointForKeypointDetection`` and requires identical
keypoints / scores / descriptors.

Architecture: VGG-style encoder 1 -> 64 -> 64 | 64 -> 64 | 128 -> 128 | 128 -> 128 (3x3 convs + ReLU, 2x2 max-pool after the first
three blocks: 1/8 resolution, 128 channels); detector head 128 -> 256 (3x3, ReLU) -> 65 (1x1), softmax over the 65 bins, dustbin
dropped, 8x8 depth-to-space, NMS (radius 4), threshold 0.005, border 4, top-k; descriptor head 128 -> 256 (3x3, ReLU) -> 256
(1x1), L2 normalise, bilinear sample at the keypoints (align_corners), L2 normalise.

State dict keys (transformers' names): ``encoder.conv_blocks.{b}.conv_{a,b}.{weight,bias}``, ``keypoint_decoder.conv_score_{a,b}.*``,
``descriptor_decoder.conv_descriptor_{a,b}.*``; conv weights are [out][in][kh][kw].
"""
from __future__ import annotations

from typing import Dict, Optional

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
KEYPOINT_THRESHOLD = 0.005
NMS_RADIUS = 4
BORDER = 4


def encoder(sd: Dict[str, Tensor], image: Tensor, taps: Optional[dict] = None) -> Tensor:
    """image (1,1,H,W) f32 in [0,1] -> (1,128,H/8,W/8)."""
    x = image
    for b in range(4):
        p = f"encoder.conv_blocks.{b}"
        x = F.relu(F.conv2d(x, sd[p + ".conv_a.weight"], sd[p + ".conv_a.bias"], padding=1))
        x = F.relu(F.conv2d(x, sd[p + ".conv_b.weight"], sd[p + ".conv_b.bias"], padding=1))
        if b < 3:
            x = F.max_pool2d(x, 2, 2)
        if taps is not None:
            taps[f"block{b}"] = x
    return x


def simple_nms(scores: Tensor, radius: int) -> Tensor:
    def max_pool(t):
        return F.max_pool2d(t, kernel_size=radius * 2 + 1, stride=1, padding=radius)

    zeros = torch.zeros_like(scores)
    max_mask = scores == max_pool(scores)
    for _ in range(2):
        supp_mask = max_pool(max_mask.float()) > 0
        supp_scores = torch.where(supp_mask, zeros, scores)
        new_max_mask = supp_scores == max_pool(supp_scores)
        max_mask = max_mask | (new_max_mask & (~supp_mask))
    return torch.where(max_mask, scores, zeros)


def pixel_scores(sd: Dict[str, Tensor], enc: Tensor) -> Tensor:
    """(1,128,h,w) -> NMS'ed score map (1, 8h, 8w)."""
    s = F.relu(F.conv2d(enc, sd["keypoint_decoder.conv_score_a.weight"], sd["keypoint_decoder.conv_score_a.bias"], padding=1))
    s = F.conv2d(s, sd["keypoint_decoder.conv_score_b.weight"], sd["keypoint_decoder.conv_score_b.bias"])
    s = F.softmax(s, 1)[:, :-1]
    b, _, h, w = s.shape
    s = s.permute(0, 2, 3, 1).reshape(b, h, w, 8, 8)
    s = s.permute(0, 1, 3, 2, 4).reshape(b, h * 8, w * 8)
    return simple_nms(s, NMS_RADIUS)


def extract_keypoints(scores: Tensor, max_keypoints: int):
    """transformers' _extract_keypoints, quirk included: the far borders are tested against 8x the map size, i.e. never."""
    _, height, width = scores.shape
    kp = torch.nonzero(scores[0] > KEYPOINT_THRESHOLD)                 # (K, 2) as (y, x), row-major order
    sc = scores[0][tuple(kp.t())]
    mask = (kp[:, 0] >= BORDER) & (kp[:, 0] < (height * 8 - BORDER)) & (kp[:, 1] >= BORDER) & (kp[:, 1] < (width * 8 - BORDER))
    kp, sc = kp[mask], sc[mask]
    if max_keypoints >= 0 and max_keypoints < len(kp):
        sc, idx = torch.topk(sc, max_keypoints, dim=0)
        kp = kp[idx]
    return torch.flip(kp, [1]).to(sc.dtype), sc                         # (x, y)


def descriptor_map(sd: Dict[str, Tensor], enc: Tensor) -> Tensor:
    d = F.conv2d(F.relu(F.conv2d(enc, sd["descriptor_decoder.conv_descriptor_a.weight"], sd["descriptor_decoder.conv_descriptor_a.bias"], padding=1)),
                 sd["descriptor_decoder.conv_descriptor_b.weight"], sd["descriptor_decoder.conv_descriptor_b.bias"])
    return F.normalize(d, p=2, dim=1)


def sample_descriptors(keypoints: Tensor, dmap: Tensor, scale: int = 8) -> Tensor:
    """keypoints (K,2) (x, y) pixels, dmap (1,256,h,w) -> (K,256)."""
    b, c, h, w = dmap.shape
    kp = keypoints[None] - scale / 2 + 0.5
    kp = kp / torch.tensor([[(w * scale - scale / 2 - 0.5), (h * scale - scale / 2 - 0.5)]]).to(kp)
    kp = kp * 2 - 1
    d = F.grid_sample(dmap, kp.view(b, 1, -1, 2), mode="bilinear", align_corners=True).reshape(b, c, -1)
    return F.normalize(d, p=2, dim=1)[0].t()


def detect_and_describe(sd: Dict[str, Tensor], gray01: Tensor, max_keypoints: int = 1024, taps: Optional[dict] = None):
    """gray01 (H,W) f32 in [0,1], H and W multiples of 8 -> (keypoints (K,2) f32 (x, y), scores (K,), descriptors (K,256))."""
    with torch.inference_mode():
        enc = encoder(sd, gray01[None, None], taps)
        scores = pixel_scores(sd, enc)
        if taps is not None:
            taps["scores"] = scores
        kp, sc = extract_keypoints(scores, max_keypoints)
        dmap = descriptor_map(sd, enc)
        if taps is not None:
            taps["dmap"] = dmap
        return kp, sc, sample_descriptors(kp, dmap)


def synthetic_state_dict(seed: int = 0) -> Dict[str, Tensor]:
    """Seeded random weights in transformers' key layout (He-style scaling so activations keep O(1) magnitude through the eight
    convolutions; the detector head is scaled so that the 65-way softmax is far from uniform and NMS has real maxima to find)."""
    g = torch.Generator().manual_seed(seed)
    sd: Dict[str, Tensor] = {}

    def conv(name, cout, cin, k, gain=1.0):
        fan_in = cin * k * k
        sd[name + ".weight"] = torch.randn(cout, cin, k, k, generator=g) * (gain * (2.0 / fan_in) ** 0.5)
        sd[name + ".bias"] = torch.randn(cout, generator=g) * 0.05

    sizes = [1, 64, 64, 128, 128]
    for b in range(4):
        conv(f"encoder.conv_blocks.{b}.conv_a", sizes[b + 1], sizes[b], 3)
        conv(f"encoder.conv_blocks.{b}.conv_b", sizes[b + 1], sizes[b + 1], 3)
    conv("keypoint_decoder.conv_score_a", 256, 128, 3)
    conv("keypoint_decoder.conv_score_b", 65, 256, 1, gain=3.0)
    conv("descriptor_decoder.conv_descriptor_a", 256, 128, 3)
    conv("descriptor_decoder.conv_descriptor_b", 256, 256, 1)
    return sd
