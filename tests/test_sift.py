"""SIFT extraction: oracle sanity on the CPU, the HIP pipeline against the oracle through the C ABI
on the GPU (bit-exact keypoints and descriptors)."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import sift as osift  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")


def blob_image(seed, h=240, w=320, n=150):
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    img = np.zeros((h, w))
    for _ in range(n):
        cx, cy, s, a = rng.uniform(0, w), rng.uniform(0, h), rng.uniform(2, 9), rng.uniform(-80, 80)
        img += a * np.exp(-((xx - cx) ** 2 + (yy - cy) ** 2) / (2 * s * s))
    return np.clip(128 + img + rng.normal(0, 2, (h, w)), 0, 255).astype(np.uint8)


# ------------------------------------------------------------------ oracle (CPU)
def test_oracle_scale_space_shapes_and_blur_preserves_constants():
    img = np.full((64, 96), 77, np.uint8)
    gauss, dog = osift.build_pyramids(img)
    assert len(gauss) == int(np.rint(np.log2(128) - 2)) + 1
    assert gauss[0][0].shape == (128, 192) and gauss[1][0].shape == (64, 96)
    for o in range(len(gauss)):
        assert all(np.abs(g - 77).max() < 1e-3 for g in gauss[o])          # Gaussian weights sum to one
        assert all(np.abs(d).max() < 1e-3 for d in dog[o])
    assert osift.detect(img) == []


def test_oracle_finds_a_blob_at_its_centre_with_the_right_scale():
    yy, xx = np.mgrid[0:128, 0:128]
    s = 6.0
    img = np.clip(40 + 160 * np.exp(-((xx - 70.3) ** 2 + (yy - 50.8) ** 2) / (2 * s * s)), 0, 255).astype(np.uint8)
    kp, size, ang, resp, octv, desc = osift.detect_and_compute(img)
    assert len(kp) >= 1
    i = int(np.argmax(resp))
    assert abs(kp[i, 0] - 70.3) < 1.0 and abs(kp[i, 1] - 50.8) < 1.0
    assert 0.6 * 2 * 1.414 * s < size[i] < 1.6 * 2 * 1.414 * s                # diameter ~ 2 sqrt(2) sigma for a Gaussian blob
    assert desc.shape == (len(kp), 128) and (desc == np.round(desc)).all() and desc.max() <= 255
    assert abs(np.linalg.norm(desc[i]) - 512) < 40


def test_oracle_helpers():
    y = np.array([0, 1, 1, 0, -1, -1, 1e-3], np.float32); x = np.array([1, 1, 0, -1, -1, 1, -5], np.float32)
    ref = np.degrees(np.arctan2(y.astype(np.float64), x.astype(np.float64))) % 360
    assert np.abs(osift.fast_atan2_deg(y, x) - ref).max() < 0.02               # OpenCV documents ~0.3 deg
    v = np.linspace(-20, 0, 1001).astype(np.float32)
    assert np.max(np.abs(osift.exp32(v) - np.exp(v.astype(np.float64))) / np.exp(v.astype(np.float64))) < 2e-6
    k = osift.gaussian_kernel(1.6)
    assert len(k) == 15 and abs(k.sum() - 1) < 1e-6 and np.array_equal(k, k[::-1])


def test_golden_sift_fixture_matches_the_oracle():
    g = np.load(os.path.join(GOLD, "sift_blobs_seed2.npz"))
    kp, size, ang, resp, octv, desc = osift.detect_and_compute(g["image"])
    assert np.array_equal(kp, g["kp"]) and np.array_equal(size, g["size"]) and np.array_equal(ang, g["angle"])
    assert np.array_equal(octv, g["octave"]) and np.array_equal(desc, g["desc"].astype(np.float32))


# ------------------------------------------------------------------ HIP pipeline vs oracle (GPU)
@pytest.fixture(scope="module")
def sift_gpu():
    import torch
    assert torch.cuda.is_available(), "these tests need an MI355X"
    from gisnav_amd.sift import SIFT
    return SIFT(max_keypoints=4096)


@pytest.mark.gpu
@pytest.mark.parametrize("seed,shape", [(0, (240, 320)), (1, (200, 333)), (2, (96, 128)), (5, (480, 640)), (6, (480, 640))])
def test_sift_keypoints_and_descriptors_bit_exact(sift_gpu, seed, shape):
    # (480, 640) is the BASELINE frame size: k_sift_tail takes octaves 4-8 there and the fused-blur tile table differs
    img = blob_image(seed, *shape, n=500 if shape[0] >= 480 else 150)
    okp, osize, oang, oresp, ooct, odesc = osift.detect_and_compute(img)
    kpt, resp, octv, desc = sift_gpu.detect_and_compute_device(img)
    k = kpt.cpu().numpy()
    assert len(k) == len(okp) > 20
    assert np.array_equal(k[:, :2].view(np.int32), okp.view(np.int32))                     # bit-exact float positions
    assert np.array_equal(k[:, 2].view(np.int32), osize.view(np.int32)) and np.array_equal(k[:, 3].view(np.int32), oang.view(np.int32))
    assert np.array_equal(resp.cpu().numpy().view(np.int32), oresp.view(np.int32)) and np.array_equal(octv.cpu().numpy(), ooct)
    assert np.array_equal(desc.cpu().numpy(), odesc)                                       # every descriptor byte


@pytest.mark.gpu
def test_sift_batch_equals_image_by_image(sift_gpu):
    """gn_sift_detect_and_compute_batch: one pass over B images gives, image for image, exactly the single-image result
    (and therefore the oracle's); also after the workspace was sized for a smaller / larger batch."""
    imgs = np.stack([blob_image(seed, 200, 264, n=150 + 40 * seed) for seed in range(5)])
    singles = [sift_gpu.detect_and_compute_device(im) for im in imgs]
    for B in (5, 2):
        kpt, resp, octv, desc, n = sift_gpu.detect_and_compute_batch_device(imgs[:B])
        assert kpt.shape[0] == B and len(n) == B
        for b in range(B):
            sk, sr, so, sd = singles[b]
            assert n[b] == len(sk) > 20
            assert np.array_equal(kpt[b, : n[b]].cpu().numpy().view(np.int32), sk.cpu().numpy().view(np.int32))
            assert np.array_equal(resp[b, : n[b]].cpu().numpy().view(np.int32), sr.cpu().numpy().view(np.int32))
            assert np.array_equal(octv[b, : n[b]].cpu().numpy(), so.cpu().numpy())
            assert np.array_equal(desc[b, : n[b]].cpu().numpy(), sd.cpu().numpy())
    odd = np.stack([blob_image(30 + b, 97, 131, n=60) for b in range(3)])             # odd width / height: unaligned rows, partial tiles
    ko, _, _, do, no = sift_gpu.detect_and_compute_batch_device(odd)
    for b in range(3):
        sk, _, _, sd = sift_gpu.detect_and_compute_device(odd[b])
        okp_b, _, _, _, _, odesc_b = osift.detect_and_compute(odd[b])
        assert no[b] == len(sk) == len(okp_b) > 5
        assert np.array_equal(ko[b, : no[b], :2].cpu().numpy(), okp_b) and np.array_equal(do[b, : no[b]].cpu().numpy(), odesc_b)
    okp, _, _, _, _, odesc = osift.detect_and_compute(imgs[3])
    kpt, _, _, desc, n = sift_gpu.detect_and_compute_batch_device(imgs)
    assert np.array_equal(kpt[3, : n[3], :2].cpu().numpy(), okp) and np.array_equal(desc[3, : n[3]].cpu().numpy(), odesc)


@pytest.mark.gpu
def test_sift_edge_cases_flat_tiny_and_overflow(sift_gpu):
    """No keypoints at all, the smallest pyramids (every octave inside the single-workgroup tail kernel, extrema search
    skipping the octaves thinner than the border), and the error path when the caller's buffers are too small."""
    from gisnav_amd.sift import SIFT
    kpt, resp, octv, desc = sift_gpu.detect_and_compute_device(np.full((64, 80), 128, np.uint8))
    assert kpt.shape == (0, 4) and desc.shape == (0, 128) and resp.shape == (0,) and octv.shape == (0,)
    kps, d = sift_gpu.detectAndCompute(np.zeros((16, 16), np.uint8), None)
    assert kps == [] and d.shape == (0, 128)
    total = 0
    for seed, shape, n in ((7, (24, 40), 20), (8, (33, 17), 10), (9, (48, 48), 30), (10, (40, 72), 40), (11, (21, 64), 25)):
        img = blob_image(seed, *shape, n=n)
        okp, osize, oang, _, ooct, odesc = osift.detect_and_compute(img)
        kpt, _, octv, desc = sift_gpu.detect_and_compute_device(img)
        k = kpt.cpu().numpy()
        assert len(k) == len(okp)
        total += len(k)
        assert np.array_equal(k[:, :2], okp) and np.array_equal(k[:, 2], osize) and np.array_equal(k[:, 3], oang)
        assert np.array_equal(octv.cpu().numpy(), ooct) and np.array_equal(desc.cpu().numpy(), odesc)
    assert total >= 15
    # more keypoints than the caller's buffers: the strongest max_keypoints by response survive, in OpenCV's order
    # (cv2 nfeatures / retainBest semantics) -- no error, and the true count is reported
    img = blob_image(0, 240, 320)
    full = osift.detect_and_compute(img)
    for cap in (16, 60):
        small = SIFT(engine=sift_gpu._eng, max_keypoints=cap)
        kpt, resp, octv, desc = small.detect_and_compute_device(img)
        okp, osize, oang, oresp, ooct, odesc = osift.retain_best(*full, cap)
        assert len(kpt) == cap == len(okp) and int(small.last_totals(1)[0]) == len(full[0]) > cap
        k = kpt.cpu().numpy()
        assert np.array_equal(k[:, :2], okp) and np.array_equal(k[:, 2], osize) and np.array_equal(k[:, 3], oang)
        assert np.array_equal(resp.cpu().numpy(), oresp) and np.array_equal(octv.cpu().numpy(), ooct)
        assert np.array_equal(desc.cpu().numpy(), odesc)
    imgs = np.stack([blob_image(s_, 200, 264, n=150 + 40 * s_) for s_ in range(3)])      # a batch where only some images overflow
    small = SIFT(engine=sift_gpu._eng, max_keypoints=120)
    kb, rb, ob, db, nb = small.detect_and_compute_batch_device(imgs)
    tot = small.last_totals(3)
    for b in range(3):
        fb = osift.detect_and_compute(imgs[b])
        okp, _, _, oresp, _, odesc = osift.retain_best(*fb, 120)
        assert tot[b] == len(fb[0]) and nb[b] == min(120, tot[b])
        assert np.array_equal(kb[b, : nb[b], :2].cpu().numpy(), okp) and np.array_equal(db[b, : nb[b]].cpu().numpy(), odesc)
    kpt, _, _, _ = sift_gpu.detect_and_compute_device(blob_image(0, 240, 320))          # the context is usable afterwards
    assert len(kpt) > 20


@pytest.mark.gpu
def test_sift_cv2_style_interface_and_golden_fixture(sift_gpu):
    g = np.load(os.path.join(GOLD, "sift_blobs_seed2.npz"))
    kps, desc = sift_gpu.detectAndCompute(g["image"], None)
    assert len(kps) == len(g["kp"])
    assert np.array_equal(np.array([k.pt for k in kps], np.float32), g["kp"])
    assert np.array_equal(np.array([k.size for k in kps], np.float32), g["size"]) and np.array_equal(np.array([k.angle for k in kps], np.float32), g["angle"])
    assert np.array_equal(desc, g["desc"].astype(np.float32))
    kps2, desc2 = sift_gpu.detectAndCompute(g["image"], None)                              # atomics reorder candidates; output must not move
    assert kps2 == kps and np.array_equal(desc2, desc)


@pytest.mark.gpu
def test_frames_to_pose_end_to_end_like_twist_node(sift_gpu):
    """TwistNode._pose from pixels: SIFT on two frames -> BFMatcher 2-NN + ratio test -> planar PnP, every stage on the
    device, against the same chain of oracles (twist_node.py:227-289)."""
    from gisnav_amd.synthetic import K_MATRIX
    from gisnav_amd.vo import twist_pose
    from oracle import bf_knn
    big = blob_image(5, 300, 400, n=260)
    ref, qry = big[20:260, 30:350], big[28:268, 41:361]                   # the camera moved by (11, 8) pixels
    okr, _, _, _, _, odr = osift.detect_and_compute(ref)
    okq, _, _, _, _, odq = osift.detect_and_compute(qry)
    kr, dr = sift_gpu.detectAndCompute(ref, None)
    kq, dq = sift_gpu.detectAndCompute(qry, None)
    pr_, pq_ = np.array([k.pt for k in kr], np.float32), np.array([k.pt for k in kq], np.float32)
    assert np.array_equal(pr_, okr) and np.array_equal(pq_, okq) and np.array_equal(dr, odr) and np.array_equal(dq, odq)
    pairs, _ = bf_knn.ratio_test(*bf_knn.knn_match2(odq, odr))
    assert len(pairs) >= 30
    shift = okq[pairs[:, 0]] - okr[pairs[:, 1]]
    assert np.abs(np.median(shift, 0) - [-11, -8]).max() < 0.2             # SIFT + ratio test found the motion
    o = bf_knn.twist_pose(K_MATRIX, okq, odq, okr, odr)
    g = twist_pose(sift_gpu._eng, K_MATRIX, pq_, dq, pr_, dr)
    assert o is not None and g is not None
    assert np.linalg.norm(g[0] - o[0]) < 1e-8 and np.linalg.norm(g[1] - o[1]) / np.linalg.norm(o[1]) < 1e-8


@pytest.mark.gpu
def test_estimate_images_equals_the_manual_chain():
    """PoseEngine.estimate_images (batched SIFT -> gn_set_active_kpts -> gn_estimate, all in HBM) against the same steps done
    image by image at the context's full padding."""
    import torch
    from gisnav_amd import _lib
    from gisnav_amd.engine import PoseEngine
    from gisnav_amd.sift import SIFT
    from gisnav_amd.synthetic import K_MATRIX
    from gisnav_amd.weights import synthetic_state_dict
    B, H, W = 3, 200, 264
    eng = PoseEngine(0, max_batch=B, max_kpts=512, precision="f16x2_bf16_attn", state_dict=synthetic_state_dict(0))
    bigs = [blob_image(40 + b, H + 12, W + 12, n=160 + 30 * b) for b in range(B)]
    tiles = np.stack([g[:H, :W] for g in bigs]); frames = np.stack([g[7:H + 7, 9:W + 9] for g in bigs])
    out, n = eng.estimate_images(frames, tiles, K_MATRIX)
    got = {k: v.cpu().numpy().copy() for k, v in out.items()}
    sift = SIFT(engine=eng, max_keypoints=512)
    desc = torch.zeros((2 * B, 512, 128), dtype=torch.float32, device=eng.device); kpt = torch.zeros((2 * B, 512, 4), dtype=torch.float32, device=eng.device)
    counts = []
    for i, im in enumerate(list(frames) + list(tiles)):
        k, _, _, d = sift.detect_and_compute_device(im)
        counts.append(len(k)); kpt[i, : len(k)] = k; desc[i, : len(k)] = d
    assert np.array_equal(n, np.array(counts, np.int32)) and min(counts) > 40 and max(counts) <= 384
    assert eng.set_active_kpts(512) == 512
    nd = torch.tensor(counts, dtype=torch.int32, device=eng.device)
    ref = eng.estimate(dict(desc_q=desc[:B], kpt_q=kpt[:B], n_q=nd[:B], desc_r=desc[B:], kpt_r=kpt[B:], n_r=nd[B:],
                            dem=torch.zeros((B, H, W), dtype=torch.uint8, device=eng.device), kpt_format=_lib.GN_KPT_XYSA), K_MATRIX)
    for k in got:
        assert np.array_equal(got[k], ref[k].cpu().numpy()), k


@pytest.mark.gpu
def test_vo_estimate_images_matches_the_oracle_chain():
    """PoseEngine.vo_estimate_images (batched SIFT -> 2-NN + ratio test -> planar PnP, all in HBM) against the chain of
    oracles pair by pair (twist_node.py:227-289)."""
    from gisnav_amd.engine import PoseEngine
    from gisnav_amd.synthetic import K_MATRIX
    from oracle import bf_knn
    B, H, W = 3, 240, 320
    eng = PoseEngine(0, max_batch=B, max_kpts=1024, precision="f32")
    bigs = [blob_image(60 + b, H + 30, W + 30, n=260) for b in range(B)]
    shifts = [(11, 8), (5, 14), (17, 3)]
    ref = np.stack([g[10:H + 10, 10:W + 10] for g in bigs])
    qry = np.stack([g[10 + dy:H + 10 + dy, 10 + dx:W + 10 + dx] for g, (dx, dy) in zip(bigs, shifts)])
    out, n = eng.vo_estimate_images(qry, ref, K_MATRIX)
    ok, R, t = out["ok"].cpu().numpy(), out["R"].cpu().numpy(), out["t"].cpu().numpy()
    for b in range(B):
        okq, _, _, _, _, odq = osift.detect_and_compute(qry[b])
        okr, _, _, _, _, odr = osift.detect_and_compute(ref[b])
        assert n[b] == len(okq) and n[B + b] == len(okr)
        o = bf_knn.twist_pose(K_MATRIX, okq, odq, okr, odr)
        assert (o is not None) == bool(ok[b])
        if o is not None:
            assert np.linalg.norm(R[b] - o[0]) < 1e-8 and np.linalg.norm(t[b] - o[1]) / np.linalg.norm(o[1]) < 1e-8
    assert ok.sum() >= 2
