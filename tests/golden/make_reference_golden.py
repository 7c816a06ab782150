"""Generate REFERENCE-DERIVED golden fixtures (tests/golden/reference/*.npz) with the reference's REAL dependencies.

This is the script that pins the oracle the day ``kornia==0.7.2`` + ``opencv-python-headless`` (+ optionally
``sift_lightglue.pth``, ``pyproj``, ``tf_transformations``) are importable in the BUILD container -- they are not today
(``import cv2|kornia`` -> ModuleNotFoundError, no wheel in /opt/wheelhouse), so no such fixture is committed and
every parity claim of this repo says "parity unpinned".  Nothing here runs on the GPU box and nothing under /root/reference is
copied: the reference's call sites are REPLAYED with their literal arguments --

  matcher     kornia.feature.LightGlueMatcher("sift", params={n_layers 9, filter_threshold 0.5, depth/width_confidence -1})
              called as ``dists, idx = m(rootsift(desc_q), rootsift(desc_r), laf_q, laf_r)``        pose_node.py:109-121,254-297
  pose        cv2.solvePnPRansac(obj, img, K, zeros((4,1)), useExtrinsicGuess=False, iterationsCount=10) + cv2.Rodrigues
                                                                                              _shared.py:89-125
  SIFT        cv2.SIFT_create().detectAndCompute(gray, None)                                   pose_node.py:122,230-232
  raster      cv2.cvtColor(BGR2GRAY), np.dstack, cv2.getRotationMatrix2D, cv2.warpAffine      stereo_node.py:229-262,292-335
  VO matcher  cv2.BFMatcher(crossCheck=False).knnMatch(q, r, k=2)                              twist_node.py:95,248-256

(The post-pose georeferencing of pose_node.py:333-381 needs pyproj + tf_transformations + ROS message types; it is pinned
analytically in tests/test_georef.py instead.)

and the outputs are stored next to the inputs.  tests/test_reference_golden.py then compares ``oracle/*`` with every fixture it
finds and SKIPS LOUDLY when there is none.

    python tests/golden/make_reference_golden.py                       # real kornia + cv2 (exit 2 with a clear message if absent)
    python tests/golden/make_reference_golden.py --weights /path/to/sift_lightglue.pth   # also the pretrained checkpoint
    python tests/golden/make_reference_golden.py --backend oracle-selftest --out /tmp/x  # plumbing test only: the ORACLE stands in
                                                                                          # for kornia / cv2; refuses tests/golden/reference

Every file records ``backend`` ("reference" | "oracle-selftest") and the versions of the packages that produced it; the
comparison test refuses "oracle-selftest" files inside tests/golden/reference/.
"""
from __future__ import annotations

import argparse
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
DEFAULT_OUT = os.path.join(HERE, "reference")

MATCH_PARAMS = {"n_layers": 9, "filter_threshold": 0.5, "depth_confidence": -1, "width_confidence": -1}   # pose_node.py:111-118


# ------------------------------------------------------------------------------------------------ shared input builders
def matcher_cases():
    """(name, weights tag, pair kwargs).  Margin-built and LOW-MARGIN synthetic weights, ragged sizes, > 1024 keypoints."""
    return [
        ("lightglue_margin_q96_r80", "margin", dict(pair_index=7, n_q=96, n_r=80)),
        ("lightglue_margin_q200_r256", "margin", dict(pair_index=8, n_q=200, n_r=256)),
        ("lightglue_lowmargin_q300_r280", "low_margin", dict(pair_index=9, n_q=300, n_r=280)),
        ("lightglue_margin_q1024_r1024", "margin", dict(pair_index=10, n_q=1024, n_r=1024)),
        ("lightglue_lowmargin_q1500_r1300", "low_margin", dict(pair_index=11, n_q=1500, n_r=1300)),
    ]


def weight_sets():
    from gisnav_amd.weights import synthetic_state_dict
    return {"margin": synthetic_state_dict(0),
            "low_margin": synthetic_state_dict(0, ffn_out_std=4.8e-3, final_scale=4.0, matchability_bias=0.0, matchability_std=0.05)}


def checkpoint_spelling(sd):
    """canonical ``transformers.{i}.self_attn.*`` -> the downloadable checkpoint's ``self_attn.{i}.*`` (kornia renames on load)."""
    out = {}
    for k, v in sd.items():
        parts = k.split(".")
        if parts[0] == "transformers":
            k = ".".join([parts[2], parts[1]] + parts[3:])
        out[k] = v
    return out


def pnp_cases():
    from gisnav_amd.synthetic import K_MATRIX, make_pair
    cases = []
    for name, seed, flat, n_out in (("pnp_dem_out60", 21, False, 60), ("pnp_flat_out60", 22, True, 60), ("pnp_dem_out0", 23, False, 0),
                                    ("pnp_flat_marginal", 24, True, 120), ("pnp_dem_k20", 25, False, 4)):
        p = make_pair(seed, flat_dem=flat)
        q = np.nonzero(p.gt_q2r >= 0)[0][:20 if name.endswith("k20") else 300]
        mq, mr = p.kp_q[q].copy(), p.kp_r[p.gt_q2r[q]]
        rs = np.random.default_rng(seed)
        mq[:n_out] = np.column_stack([rs.uniform(0, 640, n_out), rs.uniform(0, 480, n_out)]).astype(np.float32)
        cases.append((name, mq, mr, p.dem, K_MATRIX))
    # round 4: the two places where the restatement had departed from solvepnp.cpp / epnp.cpp -- exactly five points (the
    # `model_points == npoints` early return: one EPnP solve, no refinement), planar and with relief, and non-square pixels (epnp::init_points
    # re-applies the intrinsics, so the rows of M carry fx / fy): the day cv2 is available these pin both
    K_ns = K_MATRIX.copy()
    K_ns[1, 1] = 231.0
    for name, seed, flat, K, k in (("pnp_five_points_dem", 31, False, K_MATRIX, 5), ("pnp_five_points_flat", 32, True, K_MATRIX, 5),
                                   ("pnp_five_points_fy231", 33, False, K_ns, 5), ("pnp_dem_out60_fy231", 34, False, K_ns, 300),
                                   ("pnp_flat_out60_fy231", 35, True, K_ns, 300)):
        p = make_pair(seed, flat_dem=flat)
        q = np.nonzero(p.gt_q2r >= 0)[0][:k]
        mq, mr = p.kp_q[q].copy(), p.kp_r[p.gt_q2r[q]]
        if K is K_ns:      # the synthetic query points were projected with square pixels: stretch v about the principal point
            mq[:, 1] = ((mq[:, 1] - K_MATRIX[1, 2]) * (K_ns[1, 1] / K_MATRIX[1, 1]) + K_ns[1, 2]).astype(np.float32)
        if k > 5:
            rs = np.random.default_rng(seed)
            mq[:60] = np.column_stack([rs.uniform(0, 640, 60), rs.uniform(0, 480, 60)]).astype(np.float32)
        cases.append((name, mq, mr, p.dem, K))
    return cases


def images():
    """Deterministic 8-bit test images: blobs + texture at the BASELINE frame size and a small odd size."""
    out = []
    for name, (h, w), seed in (("sift_480x640", (480, 640), 2), ("sift_201x333", (201, 333), 3)):
        rs = np.random.default_rng(seed)
        yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
        img = 96.0 + 20.0 * np.sin(xx / 37.0) * np.cos(yy / 23.0)
        for _ in range(220):
            cx, cy, s, a = rs.uniform(0, w), rs.uniform(0, h), rs.uniform(1.5, 14.0), rs.uniform(-90, 90)
            img += a * np.exp(-((xx - cx) ** 2 + (yy - cy) ** 2) / (2 * s * s))
        img += rs.normal(0, 2.0, (h, w))
        out.append((name, np.clip(np.rint(img), 0, 255).astype(np.uint8)))
    return out


# ------------------------------------------------------------------------------------------------ backends
class ReferenceBackend:
    """The reference's real third-party stack.  Raises ImportError (caught in main) when it is not installed."""
    name = "reference"

    def __init__(self, weights_path=None):
        import cv2  # noqa: F401
        import kornia  # noqa: F401
        import torch  # noqa: F401
        self.cv2, self.kornia, self.torch = cv2, kornia, torch
        self.weights_path = weights_path
        self.versions = {"cv2": cv2.__version__, "kornia": kornia.__version__, "torch": torch.__version__, "numpy": np.__version__}

    # -- matcher: construct LightGlueMatcher("sift", params) exactly as pose_node.py:109-121, with OUR state dict served in
    #    place of the download (kornia's LightGlue.__init__ fetches the checkpoint through torch.hub)
    def _matcher(self, sd_np):
        torch, kornia = self.torch, self.kornia
        import kornia.feature as KF
        served = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in checkpoint_spelling(sd_np).items()}
        import kornia.feature.lightglue as klg
        saved = (torch.hub.load_state_dict_from_url, getattr(klg, "load_state_dict_from_url", None))
        torch.hub.load_state_dict_from_url = lambda *a, **k: dict(served)
        if saved[1] is not None:
            klg.load_state_dict_from_url = lambda *a, **k: dict(served)
        try:
            m = KF.LightGlueMatcher("sift", params=dict(MATCH_PARAMS)).eval()
        finally:
            torch.hub.load_state_dict_from_url = saved[0]
            if saved[1] is not None:
                klg.load_state_dict_from_url = saved[1]
        # the module must now hold exactly the served weights, under kornia's own key names
        have = m.matcher.state_dict()
        from oracle.lightglue_sift import canonical_state_dict
        want = canonical_state_dict({k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd_np.items()})
        missing = [k for k in want if k not in have and not k.startswith("token_confidence") and k != "confidence_thresholds"]
        if missing:
            raise RuntimeError(f"kornia LightGlue state_dict lacks {missing[:5]} ... : key layout differs from SURVEY.md Appendix A")
        bad = [k for k in want if k in have and not torch.equal(have[k].float().cpu(), want[k])]
        if bad:
            raise RuntimeError(f"weights not taken over verbatim for {bad[:5]}")
        return m

    def pretrained_state_dict(self):
        if not self.weights_path:
            return None
        sd = self.torch.load(self.weights_path, map_location="cpu")
        return {k: v.float().numpy() for k, v in sd.items()}

    def match(self, sd_np, p):
        torch = self.torch
        from kornia.feature import get_laf_center, laf_from_center_scale_ori
        m = self._matcher(sd_np)
        taps = {}
        hooks = []
        for i, layer in enumerate(getattr(m.matcher, "transformers", [])):
            hooks.append(layer.register_forward_hook(lambda mod, inp, out, i=i: taps.__setitem__(i, out)))
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a))  # noqa: E731
        with torch.inference_mode():   # pose_node.py:254-297, CPU tensors
            laf_q = laf_from_center_scale_ori(t(p.kp_q).unsqueeze(0), t(p.size_q)[None, :, None, None], t(p.angle_q)[None, :, None])
            laf_r = laf_from_center_scale_ori(t(p.kp_r).unsqueeze(0), t(p.size_r)[None, :, None, None], t(p.angle_r)[None, :, None])
            dq = torch.nn.functional.normalize(t(p.desc_q), dim=-1, p=1).sqrt()
            dr = torch.nn.functional.normalize(t(p.desc_r), dim=-1, p=1).sqrt()
            dists, idx = m(dq, dr, laf_q, laf_r)
            kq, kr = get_laf_center(laf_q).squeeze(), get_laf_center(laf_r).squeeze()
            mkp_q, mkp_r = kq[idx[:, 0]].numpy(), kr[idx[:, 1]].numpy()
        for h in hooks:
            h.remove()
        layer_sums = np.full((9, 2), np.nan)
        for i, out in taps.items():
            if isinstance(out, (tuple, list)) and len(out) >= 2:
                layer_sums[i] = [float(out[0].double().sum()), float(out[1].double().sum())]
        return dict(dists=dists.numpy(), idx=idx.numpy(), mkp_q=mkp_q, mkp_r=mkp_r, layer_sums=layer_sums,
                    laf_q=laf_q.numpy(), laf_r=laf_r.numpy())

    def compute_pose(self, K, mkp_q, mkp_r, dem):
        cv2 = self.cv2
        # _shared.py:95-116, literal
        x, y = np.transpose(np.floor(mkp_r).astype(int))
        obj = np.hstack((mkp_r, dem[y, x].reshape(-1, 1)))
        ok, r, t, inl = cv2.solvePnPRansac(obj, mkp_q, K, np.zeros((4, 1)), useExtrinsicGuess=False, iterationsCount=10)
        R, _ = cv2.Rodrigues(r)
        return dict(obj=obj, ok=np.array(bool(ok)), rvec=r, tvec=t, R=R, inliers=np.zeros(0, np.int32) if inl is None else np.asarray(inl).reshape(-1))

    def sift(self, gray):
        cv2 = self.cv2
        kps, desc = cv2.SIFT_create().detectAndCompute(gray, None)
        rec = np.array([[k.pt[0], k.pt[1], k.size, k.angle, k.response, k.octave] for k in kps], np.float64).reshape(-1, 6)
        capped_kps, capped_desc = cv2.SIFT_create(1024).detectAndCompute(gray, None)   # pose_node.py:108 (MAX_KEYPOINTS branch)
        capped = np.array([[k.pt[0], k.pt[1], k.size, k.angle, k.response, k.octave] for k in capped_kps], np.float64).reshape(-1, 6)
        return dict(kp=rec, desc=np.zeros((0, 128), np.float32) if desc is None else desc,
                    kp_cap1024=capped, desc_cap1024=np.zeros((0, 128), np.float32) if capped_desc is None else capped_desc)

    def stereo(self, bgr, dem, angle, crop):
        cv2 = self.cv2
        gray = cv2.cvtColor(bgr, cv2.COLOR_BGR2GRAY)
        stack = np.dstack((gray, dem))
        h, w = stack.shape[:2]
        center = (w // 2, h // 2)                                              # stereo_node.py:292-335, literal
        rot = cv2.getRotationMatrix2D(center, angle, 1.0)
        rotated = cv2.warpAffine(stack, rot, (w, h))
        dx, dy = center[0] - crop[1] // 2, center[1] - crop[0] // 2
        cropped = rotated[dy:dy + crop[0], dx:dx + crop[1]]
        inv = np.linalg.inv(np.vstack([rot, [0, 0, 1]])) @ np.array([[1, 0, dx], [0, 1, dy], [0, 0, 1]])
        return dict(gray=gray, cropped=cropped, back=inv, rot=rot)

    def knn(self, dq, dr):
        cv2 = self.cv2
        ms = cv2.BFMatcher(crossCheck=False).knnMatch(dq, dr, k=2)
        idx = np.array([[m.trainIdx for m in pair] for pair in ms], np.int32)
        dist = np.array([[m.distance for m in pair] for pair in ms], np.float32)
        return dict(idx=idx, dist=dist)


class OracleSelftestBackend:
    """PLUMBING TEST ONLY: the oracle stands in for kornia / cv2 so that this script and tests/test_reference_golden.py can be
    exercised end to end without the wheels.  Files it writes pin nothing."""
    name = "oracle-selftest"

    def __init__(self, weights_path=None):
        import torch
        self.torch = torch
        self.versions = {"torch": torch.__version__, "numpy": np.__version__, "note": "oracle stands in for kornia/cv2"}

    def pretrained_state_dict(self):
        return None

    def match(self, sd_np, p):
        torch = self.torch
        from oracle import lightglue_sift as lg
        tsd = {k: torch.from_numpy(v) for k, v in sd_np.items()}
        tq = torch.from_numpy
        taps = {}
        mq, mr, sc, idx = lg.pose_node_match(tsd, tq(p.kp_q), tq(p.desc_q), tq(p.size_q), tq(p.angle_q),
                                             tq(p.kp_r), tq(p.desc_r), tq(p.size_r), tq(p.angle_r), taps=taps)
        ls = np.array([[taps[f"layer{i}_0"].double().sum().item(), taps[f"layer{i}_1"].double().sum().item()] for i in range(9)])
        return dict(dists=sc.numpy(), idx=idx.numpy(), mkp_q=mq.numpy(), mkp_r=mr.numpy(), layer_sums=ls)

    def compute_pose(self, K, mkp_q, mkp_r, dem):
        from oracle import pnp_ransac as pr
        x, y = np.transpose(np.floor(mkp_r).astype(int))
        obj = np.hstack((mkp_r, dem[y, x].reshape(-1, 1)))
        ok, r, t, inl = pr.solve_pnp_ransac(obj, mkp_q, K, 10)
        return dict(obj=obj, ok=np.array(bool(ok)), rvec=np.asarray(r).reshape(3, 1), tvec=np.asarray(t).reshape(3, 1),
                    R=pr.rodrigues_vec2mat(r), inliers=np.asarray(inl, np.int32))

    def sift(self, gray):
        from oracle import sift as osift
        xy, size, angle, resp, octave, desc = osift.detect_and_compute(gray)
        rec = np.column_stack([xy.astype(np.float64), size, angle, resp, octave]).reshape(-1, 6)
        return dict(kp=rec, desc=desc, kp_cap1024=np.zeros((0, 6)), desc_cap1024=np.zeros((0, 128), np.float32))

    def stereo(self, bgr, dem, angle, crop):
        from oracle import stereo_warp as sw
        ref, d, back = sw.stereo_reference(bgr, dem, angle, crop)
        return dict(gray=sw.bgr2gray_u8(bgr), cropped=np.dstack((ref, d)), back=back)

    def knn(self, dq, dr):
        from oracle import bf_knn
        idx, dist = bf_knn.knn_match2(dq, dr)
        return dict(idx=idx, dist=dist)


# ------------------------------------------------------------------------------------------------ main
def generate(backend, out_dir, quick=False):
    from gisnav_amd.synthetic import K_MATRIX, make_pair
    os.makedirs(out_dir, exist_ok=True)
    meta = dict(backend=backend.name, versions=json.dumps(backend.versions))

    def save(name, **arrays):
        np.savez_compressed(os.path.join(out_dir, name + ".npz"), **meta, **arrays)
        print("wrote", name)

    wsets = weight_sets()
    pre = backend.pretrained_state_dict()
    if pre is not None:
        from gisnav_amd.weights import canonical_state_dict
        wsets["pretrained"] = canonical_state_dict(pre)
    cases = matcher_cases()
    if pre is not None:
        cases += [("lightglue_pretrained_q512_r600", "pretrained", dict(pair_index=12, n_q=512, n_r=600))]
    if quick:
        cases = cases[:3]
    for name, wtag, kw in cases:
        p = make_pair(kw["pair_index"], n_q=kw["n_q"], n_r=kw["n_r"])
        res = backend.match(wsets[wtag], p)
        pose = backend.compute_pose(K_MATRIX, res["mkp_q"], res["mkp_r"], p.dem) if len(res["idx"]) >= 15 else {}
        extra = {"weights_" + k: v for k, v in wsets[wtag].items()} if wtag == "pretrained" else {}
        save("ref_" + name, kind="matcher", weights=wtag, pair_index=kw["pair_index"], n_q=kw["n_q"], n_r=kw["n_r"],
             kp_q=p.kp_q, desc_q=p.desc_q, size_q=p.size_q, angle_q=p.angle_q, kp_r=p.kp_r, desc_r=p.desc_r, size_r=p.size_r,
             angle_r=p.angle_r, dem=p.dem, K=K_MATRIX, **res, **{"pose_" + k: v for k, v in pose.items()}, **extra)
    for name, mq, mr, dem, K in pnp_cases():
        save("ref_" + name, kind="pnp", mkp_q=mq, mkp_r=mr, dem=dem, K=K, **backend.compute_pose(K, mq, mr, dem))
    for name, img in (images()[1:] if quick else images()):
        save("ref_" + name, kind="sift", image=img, **backend.sift(img))
    rs = np.random.default_rng(3)
    bgr = rs.integers(0, 256, (300, 400, 3), dtype=np.uint8)
    dem = rs.integers(0, 256, (300, 400), dtype=np.uint8)
    for ang in (0.0, 5.0, 35.0, 90.0, 177.5, -12.25):
        save(f"ref_stereo_{ang:+07.2f}".replace(".", "p"), kind="stereo", bgr=bgr, dem=dem, angle=np.array(ang), crop=np.array([120, 160]),
             **backend.stereo(bgr, dem, ang, (120, 160)))
    p = make_pair(5, n_q=300, n_r=280)
    save("ref_knn_q300_r280", kind="knn", desc_q=p.desc_q, desc_r=p.desc_r, **backend.knn(p.desc_q, p.desc_r))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=DEFAULT_OUT)
    ap.add_argument("--weights", default=None, help="path to sift_lightglue.pth (cvg/LightGlue v0.1_arxiv)")
    ap.add_argument("--backend", default="reference", choices=["reference", "oracle-selftest"])
    ap.add_argument("--quick", action="store_true", help="fewer / smaller cases (selftest)")
    a = ap.parse_args()
    if a.backend == "oracle-selftest":
        if os.path.abspath(a.out) == os.path.abspath(DEFAULT_OUT):
            print("refusing to write oracle-selftest fixtures into tests/golden/reference/ (they pin nothing)", file=sys.stderr)
            return 3
        backend = OracleSelftestBackend()
    else:
        try:
            backend = ReferenceBackend(a.weights)
        except ImportError as e:
            print(f"REFERENCE STACK NOT IMPORTABLE ({e}).\nInstall kornia==0.7.2 and opencv-python-headless (ros/gisnav/setup.py:116-119) in the "
                  "build container and re-run; until then parity stays UNPINNED and tests/test_reference_golden.py skips.", file=sys.stderr)
            return 2
    generate(backend, a.out, a.quick)
    return 0


if __name__ == "__main__":
    sys.exit(main())
