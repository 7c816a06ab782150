"""Visual-odometry matcher (TwistNode 2-NN + ratio test): oracle KATs on the CPU, and the HIP
path against the oracle through the C ABI on the GPU."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from gisnav_amd.synthetic import K_MATRIX, make_pair  # noqa: E402
from oracle import bf_knn  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")


# ------------------------------------------------------------------ oracle known answers (CPU)
def test_oracle_knn_is_a_stable_sort_with_lower_index_first_on_ties():
    q = np.zeros((2, 128), np.float32); q[1, 0] = 10
    r = np.zeros((5, 128), np.float32)
    r[0, 0] = 3; r[1, 0] = -3; r[2, 0] = 4; r[3, 1] = 3; r[4, 0] = 10      # distances to q0: 3, 3, 4, 3, 10
    idx, dist = bf_knn.knn_match2(q, r)
    assert idx[0].tolist() == [0, 1] and dist[0].tolist() == [3.0, 3.0]     # three-way tie: train indices 0, 1 (not 3)
    assert idx[1].tolist() == [4, 2] and dist[1].tolist() == [0.0, 6.0]


def test_oracle_distances_are_exact_for_integer_descriptors():
    rng = np.random.default_rng(0)
    q = rng.integers(0, 256, (7, 128)).astype(np.float32); r = rng.integers(0, 256, (50, 128)).astype(np.float32)
    idx, dist = bf_knn.knn_match2(q, r)
    d2 = ((q[:, None, :].astype(np.int64) - r[None].astype(np.int64)) ** 2).sum(-1)
    for i in range(7):
        order = np.argsort(d2[i], kind="stable")[:2]
        assert idx[i].tolist() == order.tolist()
        assert np.array_equal(dist[i], np.sqrt(d2[i, order].astype(np.float32)))


def test_oracle_ratio_test_is_strict_and_in_double_precision():
    idx = np.array([[0, 1], [2, 3], [4, 5]], np.int32)
    dist = np.array([[7.0, 10.0], [6.9999995, 10.0], [7.0000005, 10.0]], np.float32)   # 0.7 * 10.0 == 7.0 in f64
    pairs, d = bf_knn.ratio_test(idx, dist, 0.7)
    assert pairs.tolist() == [[1, 2]] and d.tolist() == [np.float32(6.9999995)]
    with pytest.raises(ValueError):
        bf_knn.ratio_test(idx[:, :1], dist[:, :1])


def test_oracle_twist_pose_recovers_the_synthetic_motion_and_gates_on_min_matches():
    p = make_pair(3, n_q=400, n_r=400)
    got = bf_knn.twist_pose(K_MATRIX, p.kp_q, p.desc_q, p.kp_r, p.desc_r)
    assert got is not None
    assert bf_knn.twist_pose(K_MATRIX, p.kp_q[:20], p.desc_q[:20], p.kp_r, p.desc_r) is None        # < MIN_MATCHES queries
    assert bf_knn.twist_pose(K_MATRIX, p.kp_q, p.desc_q, p.kp_r[:1], p.desc_r[:1]) is None          # < 2 train descriptors


def test_golden_vo_fixture_matches_the_oracle():
    g = np.load(os.path.join(GOLD, "vo_knn_seed5_q300_r280.npz"))
    idx, dist = bf_knn.knn_match2(g["desc_q"], g["desc_r"])
    assert np.array_equal(idx, g["nn_idx"]) and np.array_equal(dist, g["nn_dist"])
    pairs, d = bf_knn.ratio_test(idx, dist)
    assert np.array_equal(pairs, g["pairs"]) and np.array_equal(d, g["pair_dist"])


# ------------------------------------------------------------------ HIP path vs oracle (GPU)
@pytest.fixture(scope="module")
def eng():
    import torch
    assert torch.cuda.is_available(), "these tests need an MI355X"
    from gisnav_amd.engine import PoseEngine
    return PoseEngine(0, max_batch=4, max_kpts=512, precision="f32")     # the VO path needs no weights


def _dev(eng, a, dt):
    import torch
    return torch.as_tensor(np.ascontiguousarray(a), dtype=dt, device=eng.device)


@pytest.mark.gpu
def test_vo_knn_and_ratio_test_bit_exact_with_ties_and_ragged_batches(eng):
    import torch
    rng = np.random.default_rng(7)
    pairs = [make_pair(10 + b, n_q=nq, n_r=nr) for b, (nq, nr) in enumerate([(300, 280), (512, 512), (31, 77), (200, 2)])]
    pairs[1].desc_r[100] = pairs[1].desc_r[37]; pairs[1].desc_r[400] = pairs[1].desc_r[37]      # exact duplicates -> ties
    pairs[2].desc_q[5] = pairs[2].desc_r[9]                                                       # a zero distance
    B, S = len(pairs), 512
    dq = np.zeros((B, S, 128), np.float32); dr = np.zeros((B, S, 128), np.float32)
    dq[:] = rng.integers(0, 256, dq.shape); dr[:] = rng.integers(0, 256, dr.shape)               # garbage beyond n must be ignored
    nq = np.array([len(p.desc_q) for p in pairs], np.int32); nr = np.array([len(p.desc_r) for p in pairs], np.int32)
    for b, p in enumerate(pairs):
        dq[b, :nq[b]] = p.desc_q; dr[b, :nr[b]] = p.desc_r
    idx, dist, n_good, nn_idx, nn_dist = eng.vo_match(_dev(eng, dq, torch.float32), _dev(eng, nq, torch.int32),
                                                      _dev(eng, dr, torch.float32), _dev(eng, nr, torch.int32), 0.7, want_knn=True)
    idx, dist, n_good, nn_idx, nn_dist = (t.cpu().numpy() for t in (idx, dist, n_good, nn_idx, nn_dist))
    for b, p in enumerate(pairs):
        oi, od = bf_knn.knn_match2(p.desc_q, p.desc_r)
        assert np.array_equal(nn_idx[b, :nq[b]], oi), b                       # bit-exact neighbours (incl. tie order)
        assert np.array_equal(nn_dist[b, :nq[b]].view(np.int32), od.view(np.int32)), b     # bit-exact f32 distances
        op, opd = bf_knn.ratio_test(oi, od)
        assert n_good[b] == len(op)
        assert np.array_equal(idx[b, :n_good[b]], op) and np.array_equal(dist[b, :n_good[b]], opd)


@pytest.mark.gpu
def test_vo_single_train_descriptor_yields_no_match(eng):
    import torch
    p = make_pair(2, n_q=64, n_r=64)
    dq, dr = p.desc_q[None], p.desc_r[None, :1]
    idx, dist, n_good, nn_idx, nn_dist = eng.vo_match(_dev(eng, dq, torch.float32), _dev(eng, np.array([64], np.int32), torch.int32),
                                                      _dev(eng, dr, torch.float32), _dev(eng, np.array([1], np.int32), torch.int32), 0.7, want_knn=True)
    assert int(n_good[0]) == 0
    assert (nn_idx[0, :64, 0] == 0).all() and (nn_idx[0, :64, 1] == -1).all()


@pytest.mark.gpu
def test_vo_pose_matches_oracle_and_min_matches_gate(eng):
    from gisnav_amd.vo import BFMatcher, twist_pose
    for seed in (3, 8):
        p = make_pair(seed, n_q=400, n_r=400)
        ref = bf_knn.twist_pose(K_MATRIX, p.kp_q, p.desc_q, p.kp_r, p.desc_r)
        got = twist_pose(eng, K_MATRIX, p.kp_q, p.desc_q, p.kp_r, p.desc_r)
        assert ref is not None and got is not None
        assert np.linalg.norm(got[0] - ref[0]) < 1e-8 and np.linalg.norm(got[1] - ref[1]) / np.linalg.norm(ref[1]) < 1e-8
    assert twist_pose(eng, K_MATRIX, p.kp_q[:20], p.desc_q[:20], p.kp_r, p.desc_r) is None
    assert twist_pose(eng, K_MATRIX, p.kp_q, p.desc_q, p.kp_r[:1], p.desc_r[:1]) is None
    bf = BFMatcher(engine=eng)                                                  # the drop-in object for twist_node.py:95
    m = bf.knnMatch(p.desc_q[:50], p.desc_r, k=2)
    oi, od = bf_knn.knn_match2(p.desc_q[:50], p.desc_r)
    assert [(a.trainIdx, b.trainIdx) for a, b in m] == [tuple(r) for r in oi.tolist()]
    assert [a.distance for a, _ in m] == [float(x) for x in od[:, 0]]
    good = bf.ratio_matches(p.desc_q, p.desc_r)
    op, _ = bf_knn.ratio_test(*bf_knn.knn_match2(p.desc_q, p.desc_r))
    assert [(g.queryIdx, g.trainIdx) for g in good] == [tuple(r) for r in op.tolist()]


@pytest.mark.gpu
def test_vo_golden_fixture_through_c_abi(eng):
    import torch
    g = np.load(os.path.join(GOLD, "vo_knn_seed5_q300_r280.npz"))
    nq, nr = len(g["desc_q"]), len(g["desc_r"])
    idx, dist, n_good, nn_idx, nn_dist = eng.vo_match(_dev(eng, g["desc_q"][None], torch.float32), _dev(eng, np.array([nq], np.int32), torch.int32),
                                                      _dev(eng, g["desc_r"][None], torch.float32), _dev(eng, np.array([nr], np.int32), torch.int32), 0.7, want_knn=True)
    assert np.array_equal(nn_idx[0, :nq].cpu().numpy(), g["nn_idx"])
    assert np.array_equal(nn_dist[0, :nq].cpu().numpy(), g["nn_dist"])
    k = int(n_good[0])
    assert np.array_equal(idx[0, :k].cpu().numpy(), g["pairs"]) and np.array_equal(dist[0, :k].cpu().numpy(), g["pair_dist"])
