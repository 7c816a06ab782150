"""Seeded synthetic LoFTR weights and image pairs for benchmarks and demos (no checkpoint can be downloaded offline; the model named by
BASELINE.json configs[1] is not in the reference tree).  Same construction, same random stream and therefore the SAME tensors as the
test oracle's generator (tests assert equality) -- but self-contained: the product never imports `oracle/`.  The calibration step of the
oracle's generator (out-convolutions made orthogonal to the mean backbone feature of a seeded scene, so that the dual-softmax has
confident mutual maxima) is applied from three stored mean vectors (`data/loftr_synth_calib_seed0.npz`, produced by
tests/golden/make_loftr_calibration.py)."""
from __future__ import annotations

import os
from typing import Dict

import numpy as np
import torch
import torch.nn.functional as F

BLOCK_DIMS = (128, 196, 256)
D_COARSE, D_FINE = 256, 128
_CALIB = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "loftr_synth_calib_seed0.npz")


def synthetic_state_dict(seed: int = 0, mlp_out_gain: float = 0.25) -> Dict[str, torch.Tensor]:
    if seed != 0:
        raise ValueError("only seed 0 has stored calibration vectors")
    g = torch.Generator().manual_seed(seed)
    sd: Dict[str, torch.Tensor] = {}

    def conv(name, cout, cin, k, gain=1.0):
        sd[name + ".weight"] = torch.randn(cout, cin, k, k, generator=g) * (gain * (2.0 / (cin * k * k)) ** 0.5)

    def bn(name, c):
        sd[name + ".weight"] = 1.0 + 0.1 * torch.randn(c, generator=g)
        sd[name + ".bias"] = 0.1 * torch.randn(c, generator=g)
        sd[name + ".running_mean"] = 0.1 * torch.randn(c, generator=g)
        sd[name + ".running_var"] = 1.0 + 0.2 * torch.rand(c, generator=g)

    def block(name, cin, cout, stride):
        conv(name + ".conv1", cout, cin, 3); bn(name + ".bn1", cout)
        conv(name + ".conv2", cout, cout, 3, gain=0.7); bn(name + ".bn2", cout)
        if stride != 1:
            conv(name + ".downsample.0", cout, cin, 1); bn(name + ".downsample.1", cout)

    d0, d1, d2 = BLOCK_DIMS
    conv("backbone.conv1", d0, 1, 7, gain=2.0); bn("backbone.bn1", d0)
    block("backbone.layer1.0", d0, d0, 1); block("backbone.layer1.1", d0, d0, 1)
    block("backbone.layer2.0", d0, d1, 2); block("backbone.layer2.1", d1, d1, 1)
    block("backbone.layer3.0", d1, d2, 2); block("backbone.layer3.1", d2, d2, 1)
    conv("backbone.layer3_outconv", d2, d2, 1, gain=2.0)
    conv("backbone.layer2_outconv", d2, d1, 1)
    conv("backbone.layer2_outconv2.0", d2, d2, 3); bn("backbone.layer2_outconv2.1", d2); conv("backbone.layer2_outconv2.3", d1, d2, 3)
    conv("backbone.layer1_outconv", d1, d0, 1)
    conv("backbone.layer1_outconv2.0", d1, d1, 3); bn("backbone.layer1_outconv2.1", d1); conv("backbone.layer1_outconv2.3", d0, d1, 3)

    def lin(name, cout, cin, gain=1.0, bias=False):
        sd[name + ".weight"] = torch.randn(cout, cin, generator=g) * (gain / cin ** 0.5)
        if bias:
            sd[name + ".bias"] = 0.05 * torch.randn(cout, generator=g)

    def layer(name, d):
        for leaf in ("q_proj", "k_proj", "v_proj", "merge"):
            lin(f"{name}.{leaf}", d, d)
        lin(f"{name}.mlp.0", 2 * d, 2 * d, gain=1.4)
        lin(f"{name}.mlp.2", d, 2 * d, gain=1.4)
        for nm in ("norm1", "norm2"):
            sd[f"{name}.{nm}.weight"] = (1.0 if nm == "norm1" else mlp_out_gain) * (1.0 + 0.1 * torch.randn(d, generator=g))
            sd[f"{name}.{nm}.bias"] = 0.02 * torch.randn(d, generator=g)

    for i in range(8):
        layer(f"loftr_coarse.layers.{i}", D_COARSE)
    for i in range(2):
        layer(f"loftr_fine.layers.{i}", D_FINE)
    lin("fine_preprocess.down_proj", D_FINE, D_COARSE, bias=True)
    lin("fine_preprocess.merge_feat", D_FINE, 2 * D_FINE, bias=True)
    z = np.load(_CALIB)
    for key in ("layer3_outconv", "layer2_outconv", "layer1_outconv"):
        m = torch.from_numpy(z[key])
        w = sd[f"backbone.{key}.weight"]
        sd[f"backbone.{key}.weight"] = (w - (w[:, :, 0, 0] @ m)[:, None, None, None] * (m / (m @ m))[None, :, None, None]).contiguous()
    return sd


def synthetic_pair(seed: int, h: int, w: int, shift=(16, 8), noise: float = 0.01):
    """Two views of one seeded textured scene (image1 = image0 shifted by `shift` pixels + independent noise): float32 (H, W) in [0, 1]."""
    rs = np.random.default_rng(seed)
    H, W = h + 64, w + 64
    base = rs.uniform(0, 1, (H // 4 + 2, W // 4 + 2)).astype(np.float32)
    big = F.interpolate(torch.from_numpy(base)[None, None], size=(H, W), mode="bicubic", align_corners=False)[0, 0]
    fine_tex = torch.from_numpy(rs.uniform(-0.15, 0.15, (H, W)).astype(np.float32))
    big = (big + fine_tex).clamp(0, 1)
    dx, dy = shift
    img0 = big[32:32 + h, 32:32 + w]
    img1 = big[32 + dy:32 + dy + h, 32 + dx:32 + dx + w]
    n0 = torch.from_numpy(rs.normal(0, noise, (h, w)).astype(np.float32))
    n1 = torch.from_numpy(rs.normal(0, noise, (h, w)).astype(np.float32))
    return (img0 + n0).clamp(0, 1).contiguous(), (img1 + n1).clamp(0, 1).contiguous()
