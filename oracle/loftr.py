"""ORACLE -- test infrastructure only, never imported by the product path.  **PARITY UNPINNED.**

CPU (torch fp32) restatement of LoFTR (Sun et al., CVPR 2021) as kornia 0.7.2 ships it (`kornia.feature.LoFTR`, default / "outdoor"
configuration) -- the matcher BASELINE.json's `north_star` and configs[1] name ("Batch-1 640x480 pair, LoFTR matcher ... fp32").
The reference tree does NOT contain LoFTR at this tag: only the word survives (`docs/vitepress/docs/glossary.md:186`,
`ros/gisnav/test/sitl/ulog_analysis/variance_estimation.ipynb:60-63`), kornia is absent from this image and no
other importable package carries the architecture (transformers' EfficientLoFTR is a different network).  So this file restates the
published architecture from kornia's module layout [EXT: kornia/feature/loftr/{loftr.py, backbone/resnet_fpn.py,
utils/position_encoding.py, loftr_module/{transformer.py, linear_attention.py, fine_preprocess.py}, utils/{coarse_matching.py,
fine_matching.py}}] and is pinned only by its own known-answer tests (tests/test_loftr.py): nothing here has been compared with the
real package.

Default configuration restated: ResNetFPN_8_2 (initial_dim 128, block_dims [128, 196, 256]); coarse transformer d_model 256, 8 heads,
['self', 'cross'] x 4, LINEAR attention (phi = elu + 1); sine position encoding with `temp_bug_fix = False` (the outdoor weights' legacy
formula); coarse matching dual-softmax, temperature 0.1, thr 0.2, border_rm 2; fine level window 5, d_model 128, 8 heads,
['self', 'cross'] x 1, concat-coarse-feature on.

State-dict keys follow kornia's module tree (`backbone.*`, `loftr_coarse.layers.{i}.*`, `loftr_fine.layers.{i}.*`,
`fine_preprocess.{down_proj,merge_feat}.*`); BatchNorm is evaluated with its running statistics (eval mode), eps 1e-5.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
BLOCK_DIMS = (128, 196, 256)
D_COARSE, D_FINE, NHEAD = 256, 128, 8
COARSE_LAYERS = ["self", "cross"] * 4
FINE_LAYERS = ["self", "cross"]
TEMPERATURE, THR, BORDER_RM = 0.1, 0.2, 2
FINE_WINDOW = 5
BN_EPS = 1e-5


# ------------------------------------------------------------------------------------------------ backbone (resnet_fpn.py)
def _bn(sd: Dict[str, Tensor], p: str, x: Tensor) -> Tensor:
    return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"], sd[p + ".weight"], sd[p + ".bias"], False, 0.0, BN_EPS)


def _basic_block(sd: Dict[str, Tensor], p: str, x: Tensor, stride: int) -> Tensor:
    """BasicBlock: relu(bn1(conv3x3(x, stride))) -> bn2(conv3x3) ; x = bn(conv1x1(x, stride)) when stride != 1 ; relu(x + y)."""
    y = F.relu(_bn(sd, p + ".bn1", F.conv2d(x, sd[p + ".conv1.weight"], None, stride=stride, padding=1)))
    y = _bn(sd, p + ".bn2", F.conv2d(y, sd[p + ".conv2.weight"], None, stride=1, padding=1))
    if stride != 1:
        x = _bn(sd, p + ".downsample.1", F.conv2d(x, sd[p + ".downsample.0.weight"], None, stride=stride))
    return F.relu(x + y)


def backbone(sd: Dict[str, Tensor], images: Tensor, taps: Optional[dict] = None) -> Tuple[Tensor, Tensor]:
    """ResNetFPN_8_2.forward: images (N,1,H,W) -> (coarse (N,256,H/8,W/8), fine (N,128,H/2,W/2))."""
    p = "backbone"
    x0 = F.relu(_bn(sd, p + ".bn1", F.conv2d(images, sd[p + ".conv1.weight"], None, stride=2, padding=3)))
    x1 = _basic_block(sd, p + ".layer1.1", _basic_block(sd, p + ".layer1.0", x0, 1), 1)        # 1/2
    x2 = _basic_block(sd, p + ".layer2.1", _basic_block(sd, p + ".layer2.0", x1, 2), 1)        # 1/4
    x3 = _basic_block(sd, p + ".layer3.1", _basic_block(sd, p + ".layer3.0", x2, 2), 1)        # 1/8
    x3_out = F.conv2d(x3, sd[p + ".layer3_outconv.weight"])
    x3_out_2x = F.interpolate(x3_out, scale_factor=2.0, mode="bilinear", align_corners=True)
    x2_out = F.conv2d(x2, sd[p + ".layer2_outconv.weight"])
    q = p + ".layer2_outconv2"
    t = F.leaky_relu(_bn(sd, q + ".1", F.conv2d(x2_out + x3_out_2x, sd[q + ".0.weight"], padding=1)))
    x2_out = F.conv2d(t, sd[q + ".3.weight"], padding=1)
    x2_out_2x = F.interpolate(x2_out, scale_factor=2.0, mode="bilinear", align_corners=True)
    x1_out = F.conv2d(x1, sd[p + ".layer1_outconv.weight"])
    q = p + ".layer1_outconv2"
    t = F.leaky_relu(_bn(sd, q + ".1", F.conv2d(x1_out + x2_out_2x, sd[q + ".0.weight"], padding=1)))
    x1_out = F.conv2d(t, sd[q + ".3.weight"], padding=1)
    if taps is not None:
        taps.update(x0=x0, x1=x1, x2=x2, x3=x3, x3_out=x3_out, x2_out=x2_out, x1_out=x1_out)
    return x3_out, x1_out


# ------------------------------------------------------------------------------------------------ position encoding
def position_encoding_sine(d_model: int, h: int, w: int, temp_bug_fix: bool = False) -> Tensor:
    """PositionEncodingSine.pe[:, :, :h, :w] as (d_model, h, w).  The legacy (temp_bug_fix = False) divisor is restated literally:
    `-math.log(10000.0) / d_model // 2` parses as floor((-ln 1e4 / d_model) / 2) = -1.0, so div_term = exp(-k), k = 0, 2, 4, ..."""
    y_position = torch.ones((h, w)).cumsum(0).float().unsqueeze(0)
    x_position = torch.ones((h, w)).cumsum(1).float().unsqueeze(0)
    k = torch.arange(0, d_model // 2, 2).float()
    if temp_bug_fix:
        div_term = torch.exp(k * (-math.log(10000.0) / (d_model // 2)))
    else:
        div_term = torch.exp(k * (-math.log(10000.0) / d_model // 2))
    div_term = div_term[:, None, None]
    pe = torch.zeros((d_model, h, w))
    pe[0::4] = torch.sin(x_position * div_term)
    pe[1::4] = torch.cos(x_position * div_term)
    pe[2::4] = torch.sin(y_position * div_term)
    pe[3::4] = torch.cos(y_position * div_term)
    return pe


# ------------------------------------------------------------------------------------------------ transformer (linear attention)
def linear_attention(q: Tensor, k: Tensor, v: Tensor, eps: float = 1e-6) -> Tensor:
    """LinearAttention.forward: q (N,L,H,D), k, v (N,S,H,D) -> (N,L,H,D).  Q = elu(q)+1, K = elu(k)+1, V = v / S;
    KV = K^T V per head; Z = 1 / (Q . sum_s K + eps); out = (Q KV) Z * S."""
    Q, K = F.elu(q) + 1, F.elu(k) + 1
    S = v.size(1)
    V = v / S
    KV = torch.einsum("nshd,nshv->nhdv", K, V)
    Z = 1 / (torch.einsum("nlhd,nhd->nlh", Q, K.sum(dim=1)) + eps)
    return torch.einsum("nlhd,nhdv,nlh->nlhv", Q, KV, Z) * S


def encoder_layer(sd: Dict[str, Tensor], p: str, x: Tensor, source: Tensor, nhead: int = NHEAD) -> Tensor:
    """LoFTREncoderLayer.forward(x, source): x + norm2(mlp(cat[x, norm1(merge(attn(q(x), k(src), v(src))))]))."""
    n, L, d = x.shape
    dim = d // nhead
    q = F.linear(x, sd[p + ".q_proj.weight"]).view(n, -1, nhead, dim)
    k = F.linear(source, sd[p + ".k_proj.weight"]).view(n, -1, nhead, dim)
    v = F.linear(source, sd[p + ".v_proj.weight"]).view(n, -1, nhead, dim)
    msg = linear_attention(q, k, v).reshape(n, -1, d)
    msg = F.layer_norm(F.linear(msg, sd[p + ".merge.weight"]), (d,), sd[p + ".norm1.weight"], sd[p + ".norm1.bias"], 1e-5)
    msg = F.linear(F.relu(F.linear(torch.cat([x, msg], dim=2), sd[p + ".mlp.0.weight"])), sd[p + ".mlp.2.weight"])
    msg = F.layer_norm(msg, (d,), sd[p + ".norm2.weight"], sd[p + ".norm2.bias"], 1e-5)
    return x + msg


def local_feature_transformer(sd: Dict[str, Tensor], p: str, names: List[str], feat0: Tensor, feat1: Tensor, taps: Optional[dict] = None):
    """LocalFeatureTransformer.forward: 'self' updates both sides independently; 'cross' updates feat0 from feat1 and THEN feat1 from the
    UPDATED feat0 (the sequential order of the published code)."""
    for i, name in enumerate(names):
        lp = f"{p}.layers.{i}"
        if name == "self":
            feat0 = encoder_layer(sd, lp, feat0, feat0)
            feat1 = encoder_layer(sd, lp, feat1, feat1)
        else:
            feat0 = encoder_layer(sd, lp, feat0, feat1)
            feat1 = encoder_layer(sd, lp, feat1, feat0)
        if taps is not None:
            taps[f"{p}.{i}"] = (feat0, feat1)
    return feat0, feat1


# ------------------------------------------------------------------------------------------------ coarse matching
def mask_border(m: Tensor, b: int, v: bool) -> None:
    """utils/coarse_matching.py mask_border on (N, H0, W0, H1, W1)."""
    if b <= 0:
        return
    m[:, :b] = v
    m[:, :, :b] = v
    m[:, :, :, :b] = v
    m[:, :, :, :, :b] = v
    m[:, -b:] = v
    m[:, :, -b:] = v
    m[:, :, :, -b:] = v
    m[:, :, :, :, -b:] = v


def coarse_matching(feat_c0: Tensor, feat_c1: Tensor, hw0_c, hw1_c, scale: int = 8, thr: float = THR, border_rm: int = BORDER_RM,
                    temperature: float = TEMPERATURE, taps: Optional[dict] = None):
    """CoarseMatching.forward (dual_softmax) + get_coarse_match: returns (b_ids, i_ids, j_ids, mconf, mkpts0_c, mkpts1_c)."""
    c = feat_c0.shape[-1]
    f0, f1 = feat_c0 / c ** 0.5, feat_c1 / c ** 0.5
    sim = torch.einsum("nlc,nsc->nls", f0, f1) / temperature
    conf = F.softmax(sim, 1) * F.softmax(sim, 2)
    if taps is not None:
        taps["conf_matrix"] = conf
    n = conf.shape[0]
    mask = (conf > thr).view(n, hw0_c[0], hw0_c[1], hw1_c[0], hw1_c[1]).clone()
    mask_border(mask, border_rm, False)
    mask = mask.view(n, hw0_c[0] * hw0_c[1], hw1_c[0] * hw1_c[1])
    mask = mask * (conf == conf.max(dim=2, keepdim=True)[0]) * (conf == conf.max(dim=1, keepdim=True)[0])
    mask_v, all_j = mask.max(dim=2)
    b_ids, i_ids = torch.where(mask_v)
    j_ids = all_j[b_ids, i_ids]
    mconf = conf[b_ids, i_ids, j_ids]
    mk0 = torch.stack([i_ids % hw0_c[1], i_ids // hw0_c[1]], dim=1) * scale
    mk1 = torch.stack([j_ids % hw1_c[1], j_ids // hw1_c[1]], dim=1) * scale
    return b_ids, i_ids, j_ids, mconf, mk0.float(), mk1.float()


# ------------------------------------------------------------------------------------------------ fine level
def fine_preprocess(sd: Dict[str, Tensor], feat_f0: Tensor, feat_f1: Tensor, feat_c0: Tensor, feat_c1: Tensor, b_ids, i_ids, j_ids):
    """FinePreprocess.forward: 5x5 windows of the 1/2-resolution maps around every coarse match (unfold, stride 4, padding 2), merged with
    the coarse features of the match (down_proj, merge_feat).  Returns (M, 25, 128) x 2."""
    W = FINE_WINDOW
    stride = 4   # hw0_f[0] // hw0_c[0]
    M = len(b_ids)
    if M == 0:
        z = torch.empty(0, W * W, D_FINE)
        return z, z
    c = feat_f0.shape[1]
    u0 = F.unfold(feat_f0, kernel_size=(W, W), stride=stride, padding=W // 2).view(feat_f0.shape[0], c, W * W, -1).permute(0, 3, 2, 1)
    u1 = F.unfold(feat_f1, kernel_size=(W, W), stride=stride, padding=W // 2).view(feat_f1.shape[0], c, W * W, -1).permute(0, 3, 2, 1)
    u0, u1 = u0[b_ids, i_ids], u1[b_ids, j_ids]                                         # (M, 25, 128)
    c_win = F.linear(torch.cat([feat_c0[b_ids, i_ids], feat_c1[b_ids, j_ids]], 0), sd["fine_preprocess.down_proj.weight"], sd["fine_preprocess.down_proj.bias"])
    cf = F.linear(torch.cat([torch.cat([u0, u1], 0), c_win[:, None].expand(-1, W * W, -1)], -1),
                  sd["fine_preprocess.merge_feat.weight"], sd["fine_preprocess.merge_feat.bias"])
    return cf[:M], cf[M:]


def spatial_expectation2d_normalized(heat: Tensor) -> Tensor:
    """kornia.geometry.subpix.dsnt.spatial_expectation2d(heat[None], normalized_coordinates=True)[0]: (M, W, W) -> (M, 2) (x, y) in [-1, 1]."""
    m, h, w = heat.shape
    xs = torch.linspace(-1, 1, w)
    ys = torch.linspace(-1, 1, h)
    ex = (heat * xs[None, None, :]).sum((1, 2))
    ey = (heat * ys[None, :, None]).sum((1, 2))
    return torch.stack([ex, ey], -1)


def fine_matching(feat_f0: Tensor, feat_f1: Tensor, mkpts0_c: Tensor, mkpts1_c: Tensor, scale_f: float = 2.0):
    """FineMatching.forward: correlate the centre feature of window 0 with window 1, softmax(1/sqrt(C)) heatmap, expectation;
    mkpts0_f = mkpts0_c, mkpts1_f = mkpts1_c + expectation * (W // 2) * scale."""
    M, WW, C = feat_f0.shape
    W = int(math.sqrt(WW))
    if M == 0:
        return mkpts0_c, mkpts1_c
    picked = feat_f0[:, WW // 2, :]
    sim = torch.einsum("mc,mrc->mr", picked, feat_f1)
    heat = torch.softmax(sim / C ** 0.5, dim=1).view(-1, W, W)
    coords = spatial_expectation2d_normalized(heat)
    return mkpts0_c, mkpts1_c + coords * (W // 2) * scale_f


# ------------------------------------------------------------------------------------------------ whole model
def loftr_forward(sd: Dict[str, Tensor], image0: Tensor, image1: Tensor, taps: Optional[dict] = None, fine: bool = True):
    """LoFTR.forward on one pair of (H, W) float images in [0, 1] of EQUAL size, H and W multiples of 8.  Returns kornia's output dict:
    keypoints0 / keypoints1 (M, 2) (x, y) pixels, confidence (M,), batch_indexes (M,) -- plus the coarse-level ids."""
    with torch.inference_mode():
        h, w = image0.shape
        feats_c, feats_f = backbone(sd, torch.stack([image0, image1])[:, None], taps)
        hc, wc = h // 8, w // 8
        pe = position_encoding_sine(D_COARSE, hc, wc)
        fc = (feats_c + pe[None]).flatten(2).transpose(1, 2)                           # (2, hc*wc, 256)
        if taps is not None:
            taps["coarse_in"] = fc
        f0, f1 = local_feature_transformer(sd, "loftr_coarse", COARSE_LAYERS, fc[:1], fc[1:], taps)
        b_ids, i_ids, j_ids, mconf, mk0, mk1 = coarse_matching(f0, f1, (hc, wc), (hc, wc), 8, taps=taps)
        out = dict(i_ids=i_ids, j_ids=j_ids, confidence=mconf, batch_indexes=b_ids, keypoints0_c=mk0, keypoints1_c=mk1)
        if fine:
            ff0, ff1 = fine_preprocess(sd, feats_f[:1], feats_f[1:], f0, f1, b_ids, i_ids, j_ids)
            if len(b_ids):
                ff0, ff1 = local_feature_transformer(sd, "loftr_fine", FINE_LAYERS, ff0, ff1)
            if taps is not None:
                taps["fine_windows"] = (ff0, ff1)
            mk0, mk1 = fine_matching(ff0, ff1, mk0, mk1)
        out.update(keypoints0=mk0, keypoints1=mk1)
        return out


# ------------------------------------------------------------------------------------------------ seeded weights
def synthetic_state_dict(seed: int = 0, mlp_out_gain: float = 0.25) -> Dict[str, Tensor]:
    """Seeded random weights in kornia's key layout.  He-scaled convolutions and O(1) BatchNorm statistics keep the activations O(1)
    through the backbone; the transformer's second MLP matrix is scaled down (`mlp_out_gain`) so that every layer perturbs rather than
    replaces the stream -- corresponding cells of two views of one scene then keep similar coarse features and the dual-softmax has
    confident mutual maxima to find (random weights, so WHAT is matched means nothing; that decisions have margins is what the parity
    tests need)."""
    g = torch.Generator().manual_seed(seed)
    sd: Dict[str, Tensor] = {}

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

    for i in range(len(COARSE_LAYERS)):
        layer(f"loftr_coarse.layers.{i}", D_COARSE)
    for i in range(len(FINE_LAYERS)):
        layer(f"loftr_fine.layers.{i}", D_FINE)
    lin("fine_preprocess.down_proj", D_FINE, D_COARSE, bias=True)
    lin("fine_preprocess.merge_feat", D_FINE, 2 * D_FINE, bias=True)
    # Calibration (what training does for the real weights): the ReLU features entering the 1x1 out-convolutions have a large common
    # mean; left in, every coarse descriptor shares one dominant direction and all rows of the similarity matrix prefer the same
    # column.  The out-convolutions are made orthogonal to the mean feature of a seeded calibration scene.
    with torch.inference_mode():
        img, _ = synthetic_pair(12345, 96, 128)
        taps: dict = {}
        backbone(sd, img[None, None], taps)
        for key, tap in (("backbone.layer3_outconv", "x3"), ("backbone.layer2_outconv", "x2"), ("backbone.layer1_outconv", "x1")):
            m = taps[tap].mean((0, 2, 3))
            w = sd[key + ".weight"]
            sd[key + ".weight"] = (w - (w[:, :, 0, 0] @ m)[:, None, None, None] * (m / (m @ m))[None, :, None, None]).contiguous()
    return sd


def synthetic_pair(seed: int, h: int, w: int, shift=(16, 8), noise: float = 0.01):
    """Two views of one seeded textured scene: image1 is image0 shifted by `shift` = (dx, dy) pixels (multiples of 8 make the coarse
    cells correspond exactly) plus a little independent noise.  Returns float32 images in [0, 1], (H, W) each."""
    import numpy as np
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
