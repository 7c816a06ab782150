"""LightGlue("sift") weights in kornia 0.7.2's state-dict layout.

The pretrained ``sift_lightglue.pth`` that kornia downloads at first use (implicit in
ros/gisnav/gisnav/core/pose_node.py:109-121) is not available offline, so this module
provides (a) a seeded synthetic generator with exactly the checkpoint's key names and
shapes -- real weights drop in unchanged -- and (b) the loader-side key canonicalisation.

Synthetic weights are built so that the network does real work on realistic value ranges
AND still produces true matches (otherwise the PnP stage would never run in the benchmark):
``input_proj`` has orthonormal columns, the last FFN linear of every block is small (the
residual stream stays dominated by the projected RootSIFT descriptor, perturbed ~5 % per
block), q/k projections are scaled so attention logits have O(1) spread, ``final_proj`` is
a scaled orthogonal matrix and ``matchability`` is biased positive.  Every GEMM, softmax,
LayerNorm and GELU of the real model is executed on dense non-trivial data.
"""
from __future__ import annotations

from typing import Dict

import numpy as np

N_LAYERS = 9
NUM_HEADS = 4
DESC_DIM = 256
INPUT_DIM = 128
HEAD_DIM = 64


def _uniform(rng, shape, bound):
    return rng.uniform(-bound, bound, size=shape).astype(np.float32)


def _orthogonal(rng, n):
    q, r = np.linalg.qr(rng.normal(size=(n, n)))
    return (q * np.sign(np.diag(r))).astype(np.float64)


def _ffn(rng, sd, p, d, ffn_out_std):
    sd[p + ".ffn.0.weight"] = _uniform(rng, (2 * d, 2 * d), (2 * d) ** -0.5)
    sd[p + ".ffn.0.bias"] = _uniform(rng, (2 * d,), (2 * d) ** -0.5)
    sd[p + ".ffn.1.weight"] = (1.0 + 0.1 * rng.normal(size=2 * d)).astype(np.float32)
    sd[p + ".ffn.1.bias"] = (0.1 * rng.normal(size=2 * d)).astype(np.float32)
    sd[p + ".ffn.3.weight"] = (ffn_out_std * rng.normal(size=(d, 2 * d))).astype(np.float32)
    sd[p + ".ffn.3.bias"] = (ffn_out_std * rng.normal(size=d)).astype(np.float32)


def synthetic_state_dict(seed: int = 0, n_layers: int = N_LAYERS, qk_gain: float = 30.0,
                         ffn_out_std: float = 2.4e-4, final_scale: float = 28.0,
                         identity_blocks: bool = False, matchability_bias: float = 10.0,
                         matchability_std: float = 0.01, feature: str = "sift") -> Dict[str, np.ndarray]:
    """Seeded weights, kornia key layout (``transformers.{i}.self_attn.*`` ...).

    ``identity_blocks=True`` zeroes every ``ffn.3`` so each block is the identity on the
    residual stream: the matcher then reduces to mutual-NN of RootSIFT descriptors
    (SURVEY.md 8(c) KAT 2).

    The defaults build a wide decision margin (the match is dominated by the projected descriptor).  A LOW-MARGIN set --
    e.g. ``ffn_out_std=4.8e-3`` (each block perturbs the stream ~100 %), ``final_scale=4``, ``matchability_bias=0``,
    ``matchability_std=0.05`` -- makes near-ties common, which is what the precision-mode mismatch tests need.
    """
    rng = np.random.default_rng(seed)
    d = DESC_DIM
    sd: Dict[str, np.ndarray] = {}
    ip_w = _orthogonal(rng, d)[:, :INPUT_DIM].astype(np.float32)  # (256,128), W^T W = I
    ip_b = (0.01 * rng.normal(size=d)).astype(np.float32)
    wr = rng.normal(size=(HEAD_DIM // 2, 4)) * np.array([3.0, 3.0, 0.05, 0.3])
    if feature == "sift":
        sd["input_proj.weight"], sd["input_proj.bias"] = ip_w, ip_b
        sd["posenc.Wr.weight"] = wr.astype(np.float32)  # input order (x, y, scale, ori_rad)
    else:   # kornia LightGlue(features="superpoint"): input_dim == descriptor_dim (no input_proj), add_scale_ori False: Wr on (x, y) only
        sd["posenc.Wr.weight"] = np.ascontiguousarray(wr[:, :2]).astype(np.float32)
    b = d ** -0.5
    for i in range(n_layers):
        p = f"transformers.{i}.self_attn"
        w = _uniform(rng, (3 * d, d), b)
        bias = _uniform(rng, (3 * d,), b)
        # flat output index = h*192 + dd*3 + s, s in {0:q, 1:k, 2:v}  (unflatten(-1,(4,64,3)))
        s_idx = np.arange(3 * d) % 3
        gain = np.where(s_idx < 2, qk_gain, 4.0).astype(np.float32)
        sd[p + ".Wqkv.weight"] = w * gain[:, None]
        sd[p + ".Wqkv.bias"] = bias * np.where(s_idx < 2, 1.0, 1.0).astype(np.float32)
        sd[p + ".out_proj.weight"] = _uniform(rng, (d, d), b)
        sd[p + ".out_proj.bias"] = _uniform(rng, (d,), b)
        _ffn(rng, sd, p, d, 0.0 if identity_blocks else ffn_out_std)
        p = f"transformers.{i}.cross_attn"
        sd[p + ".to_qk.weight"] = _uniform(rng, (d, d), b) * np.float32(qk_gain)
        sd[p + ".to_qk.bias"] = _uniform(rng, (d,), b)
        sd[p + ".to_v.weight"] = _uniform(rng, (d, d), b) * np.float32(4.0)
        sd[p + ".to_v.bias"] = _uniform(rng, (d,), b)
        sd[p + ".to_out.weight"] = _uniform(rng, (d, d), b)
        sd[p + ".to_out.bias"] = _uniform(rng, (d,), b)
        _ffn(rng, sd, p, d, 0.0 if identity_blocks else ffn_out_std)
    for i in range(n_layers):
        p = f"log_assignment.{i}"
        sd[p + ".final_proj.weight"] = (final_scale * _orthogonal(rng, d)).astype(np.float32)
        sd[p + ".final_proj.bias"] = (0.01 * rng.normal(size=d)).astype(np.float32)
        sd[p + ".matchability.weight"] = (matchability_std * rng.normal(size=(1, d))).astype(np.float32)
        sd[p + ".matchability.bias"] = np.array([matchability_bias], np.float32)
    for i in range(n_layers - 1):
        p = f"token_confidence.{i}.token.0"  # unused when depth/width_confidence = -1
        sd[p + ".weight"] = _uniform(rng, (1, d), b)
        sd[p + ".bias"] = _uniform(rng, (1,), b)
    sd["confidence_thresholds"] = np.clip(0.8 + 0.1 * np.exp(-4.0 * np.arange(n_layers) / n_layers), 0, 1).astype(np.float32)
    return sd


def default_init_state_dict(seed: int = 0, n_layers: int = N_LAYERS, feature: str = "sift") -> Dict[str, np.ndarray]:
    """A third weight family: every tensor as PyTorch's own constructors would leave it -- `nn.Linear`: weight and bias
    ~ U(-1 / sqrt(in), 1 / sqrt(in)) (kaiming_uniform with a = sqrt(5)); `nn.LayerNorm`: weight 1, bias 0 -- i.e. what
    `kornia.feature.LightGlue("sift")` holds before `sift_lightglue.pth` is loaded (pose_node.py:109-121).  Nothing is hand-shrunk: each block
    rewrites the residual stream completely, activations take whatever range nine un-trained layers give them, and the assignment matrix
    is nearly flat -- with filter_threshold 0 every mutual arg-max is a match and most of them have a small margin.  A look at a realistic
    dynamic range through the fp16 split, the fp16 attention and the guards; not a model that finds true correspondences."""
    rng = np.random.default_rng(50_000 + seed)
    d = DESC_DIM
    sd: Dict[str, np.ndarray] = {}

    def linear(name, out, inp):
        b = inp ** -0.5
        sd[name + ".weight"] = _uniform(rng, (out, inp), b)
        sd[name + ".bias"] = _uniform(rng, (out,), b)

    if feature == "sift":
        linear("input_proj", d, INPUT_DIM)
        sd["posenc.Wr.weight"] = _uniform(rng, (HEAD_DIM // 2, 4), 4 ** -0.5)
    else:
        sd["posenc.Wr.weight"] = _uniform(rng, (HEAD_DIM // 2, 2), 2 ** -0.5)
    for i in range(n_layers):
        for blk, lins in (("self_attn", (("Wqkv", 3 * d, d), ("out_proj", d, d))), ("cross_attn", (("to_qk", d, d), ("to_v", d, d), ("to_out", d, d)))):
            p = f"transformers.{i}.{blk}"
            for name, o, k in lins:
                linear(f"{p}.{name}", o, k)
            linear(p + ".ffn.0", 2 * d, 2 * d)
            sd[p + ".ffn.1.weight"] = np.ones(2 * d, np.float32)
            sd[p + ".ffn.1.bias"] = np.zeros(2 * d, np.float32)
            linear(p + ".ffn.3", d, 2 * d)
    for i in range(n_layers):
        linear(f"log_assignment.{i}.final_proj", d, d)
        linear(f"log_assignment.{i}.matchability", 1, d)
    for i in range(n_layers - 1):
        linear(f"token_confidence.{i}.token.0", 1, d)
    sd["confidence_thresholds"] = np.clip(0.8 + 0.1 * np.exp(-4.0 * np.arange(n_layers) / n_layers), 0, 1).astype(np.float32)
    return sd


def canonical_key(k: str) -> str:
    """Checkpoint spelling ``self_attn.{i}.*`` -> kornia's ``transformers.{i}.self_attn.*``."""
    for kind in ("self_attn", "cross_attn"):
        if k.startswith(kind + "."):
            idx, tail = k[len(kind) + 1:].split(".", 1)
            return f"transformers.{idx}.{kind}.{tail}"
    return k


def canonical_state_dict(sd) -> Dict[str, np.ndarray]:
    out = {}
    for k, v in sd.items():
        if hasattr(v, "detach"):
            v = v.detach().cpu().numpy()
        out[canonical_key(k)] = np.ascontiguousarray(v, dtype=np.float32)
    return out


def expected_shapes(n_layers: int = N_LAYERS, feature: str = "sift") -> Dict[str, tuple]:
    """Shape table of SURVEY.md Appendix A, used by the loader to validate a checkpoint."""
    d = DESC_DIM
    shp = {"input_proj.weight": (d, INPUT_DIM), "input_proj.bias": (d,), "posenc.Wr.weight": (HEAD_DIM // 2, 4)} if feature == "sift" \
        else {"posenc.Wr.weight": (HEAD_DIM // 2, 2)}
    for i in range(n_layers):
        for blk, lin in (("self_attn", (("Wqkv", 3 * d, d), ("out_proj", d, d))),
                         ("cross_attn", (("to_qk", d, d), ("to_v", d, d), ("to_out", d, d)))):
            p = f"transformers.{i}.{blk}"
            for name, o, k in lin:
                shp[f"{p}.{name}.weight"] = (o, k)
                shp[f"{p}.{name}.bias"] = (o,)
            shp[p + ".ffn.0.weight"] = (2 * d, 2 * d)
            shp[p + ".ffn.0.bias"] = (2 * d,)
            shp[p + ".ffn.1.weight"] = (2 * d,)
            shp[p + ".ffn.1.bias"] = (2 * d,)
            shp[p + ".ffn.3.weight"] = (d, 2 * d)
            shp[p + ".ffn.3.bias"] = (d,)
        shp[f"log_assignment.{i}.final_proj.weight"] = (d, d)
        shp[f"log_assignment.{i}.final_proj.bias"] = (d,)
        shp[f"log_assignment.{i}.matchability.weight"] = (1, d)
        shp[f"log_assignment.{i}.matchability.bias"] = (1,)
    return shp
