"""CPU tests of the margin certificate's LOGIC (the GPU side is tests/test_gpu_round6.py) and of the oracle's half-SDPA mode.

The certificate (include/gisnav_amd.h, gn_set_certify) claims: if two arithmetics' assignment matrices P, Q differ by at most eps on the deciding
entries, and on P (a) every row whose best score is >= L - eps leads its runner-up by more than 2 eps, (b) the same for every column, (c) no row's
best score is within eps of L = log(filter_threshold), then `filter_matches` returns the same list on P and Q (kornia's mutual arg-max + threshold,
pose_node.py:285-297 consumes the list).  Checked here on the oracle's own two arithmetics -- fp32 attention and the emulated half-precision SDPA of the
reference's CUDA branch (pose_node.py:81, 108-121) -- which is also VERDICT r5 item 2(d): how many indices the reference's own CPU -> CUDA move flips.
"""
import math

import numpy as np
import pytest
import torch

from conftest import oracle_match
from gisnav_amd.synthetic import make_pair
from gisnav_amd.weights import default_init_state_dict, synthetic_state_dict
from oracle import lightglue_sift as lg

LOW_MARGIN = dict(ffn_out_std=4.8e-3, final_scale=4.0, matchability_bias=0.0, matchability_std=0.05)


def certificate(P: np.ndarray, th: float, eps: float) -> bool:
    """True = certified.  The numpy statement of what k_head_fused's last workgroup evaluates (csrc/gn_match_head.hip)."""
    L = math.log(th) if th > 0 else -math.inf
    for M in (P, P.T):                                   # rows (a), columns (b)
        top2 = -np.sort(-M, axis=1)[:, :2]
        near = top2[:, 0] >= L - eps
        if np.any(near & ~(top2[:, 0] - top2[:, 1] > 2 * eps)):
            return False
    best = P.max(1)
    return not np.any(np.abs(best - L) <= eps)           # (c)


def _both(sd, p, th):
    taps, taps2 = {}, {}
    r = oracle_match(sd, p, taps=taps, filter_threshold=th)
    with lg.attention_mode("half_sdpa"):
        r2 = oracle_match(sd, p, taps=taps2, filter_threshold=th)
    assert lg.ATTENTION_MODE == "fp32"
    P, Q = taps["scores"][0, :-1, :-1].numpy(), taps2["scores"][0, :-1, :-1].numpy()
    a, b = {tuple(x) for x in r[3].tolist()}, {tuple(x) for x in r2[3].tolist()}
    return P, Q, a, b


@pytest.mark.parametrize("family,th", [("low", 0.0), ("low", 0.2), ("default", 0.0), ("bench", 0.5)])
def test_certified_pairs_have_identical_match_lists_in_both_arithmetics(family, th):
    torch.set_num_threads(8)
    sdn = {"low": lambda: synthetic_state_dict(0, **LOW_MARGIN), "default": lambda: default_init_state_dict(0), "bench": lambda: synthetic_state_dict(0)}[family]()
    sd = {k: torch.from_numpy(v) for k, v in sdn.items()}
    certified = flipped_pairs = flips = matches = 0
    for i in range(10):
        p = make_pair(3000 + i, n_q=200 - 7 * i, n_r=190)
        P, Q, a, b = _both(sd, p, th)
        eps = float(np.abs(P - Q).max())                 # the exact bound for this pair (the product calibrates it on a sample, x safety 4)
        ok = certificate(Q, th, eps)                     # certify the HALF arithmetic's matrix, as the fast GPU mode certifies its own
        certified += ok
        flips += len(a ^ b); flipped_pairs += bool(a ^ b); matches += len(a)
        if ok:
            assert a == b, (family, th, i, sorted(a ^ b))
    print(f"{family} th={th}: {matches} matches, CPU->half-SDPA flips {flips} in {flipped_pairs} pairs, {certified}/10 pairs certified at their exact eps")
    if family == "bench":
        assert certified == 10 and flips == 0            # margin-built weights: every decision is far from flipping


def test_threshold_half_and_above_implies_the_gap_conditions():
    """For filter_threshold >= 0.5 (PoseNode's value) conditions (a) and (b) follow from (c): an entry whose score exceeds 0.5 e^eps leaves less than
    0.5 e^-eps for the rest of its row and column.  Numerically: on random double-softmax matrices, (c) alone decides the certificate."""
    rng = np.random.default_rng(0)
    L = math.log(0.5)
    for trial in range(200):
        n, m = int(rng.integers(2, 40)), int(rng.integers(2, 40))
        sim = rng.normal(size=(n, m)) * rng.uniform(0.5, 12.0)
        z0, z1 = rng.normal(size=n) * 3 + 2, rng.normal(size=m) * 3 + 2
        P = lg.sigmoid_log_double_softmax(torch.from_numpy(sim)[None], torch.from_numpy(z0)[None, :, None], torch.from_numpy(z1)[None, :, None])[0, :-1, :-1].numpy()
        eps = float(rng.uniform(1e-6, 0.05))
        c_only = not np.any(np.abs(P.max(1) - L) <= eps)
        assert certificate(P, 0.5, eps) == c_only, trial


def test_half_sdpa_mode_rounds_like_the_cuda_branch_and_is_off_by_default():
    torch.manual_seed(0)
    q, k, v = (torch.randn(1, 4, 50, 64) for _ in range(3))
    ref = lg.attention(q, k, v)
    with lg.attention_mode("half_sdpa"):
        got = lg.attention(q, k, v)
    assert lg.ATTENTION_MODE == "fp32"
    assert torch.equal(got, got.half().float())                         # the output went through fp16
    assert 1e-5 < float((got - ref).abs().max()) < 5e-3                 # and differs from fp32 by half-precision rounding, no more
    if hasattr(torch.nn.functional, "scaled_dot_product_attention"):    # torch's own CPU half SDPA (math path) agrees to half rounding
        sd = torch.nn.functional.scaled_dot_product_attention(q.half(), k.half(), v.half()).float()
        assert float((got - sd).abs().max()) < 4e-3
