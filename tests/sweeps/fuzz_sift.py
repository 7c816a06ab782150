"""Developer tool: the SIFT extractor against its oracle over image sizes / contents the test-suite does not hold: odd sizes, very wide / tall, high
noise (many keypoints), strong edges, saturated regions.  Every keypoint field and descriptor byte must be identical.   python tests/sweeps/fuzz_sift.py"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import sift as osift  # noqa: E402   (checker, as in tests/)
from gisnav_amd.sift import SIFT  # noqa: E402


def blobs(seed, h, w, n, noise=2.0):
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    img = np.zeros((h, w))
    for _ in range(n):
        cx, cy, s, a = rng.uniform(0, w), rng.uniform(0, h), rng.uniform(1.5, 12), rng.uniform(-90, 90)
        img += a * np.exp(-((xx - cx) ** 2 + (yy - cy) ** 2) / (2 * s * s))
    return np.clip(128 + img + rng.normal(0, noise, (h, w)), 0, 255).astype(np.uint8)


def checker(seed, h, w):
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    t = ((xx // 13 + yy // 17) % 2) * 200 + 20
    return np.clip(t + rng.normal(0, 3, (h, w)), 0, 255).astype(np.uint8)


cases = [("blobs 97x131", blobs(1, 97, 131, 60)), ("blobs 333x77 (tall)", blobs(2, 333, 77, 80)), ("blobs 61x509 (wide)", blobs(3, 61, 509, 80)),
         ("noisy 200x264", blobs(4, 200, 264, 100, noise=12.0)), ("checkerboard 240x320", checker(5, 240, 320)),
         ("saturated 150x210", np.clip(blobs(6, 150, 210, 120).astype(int) * 3 - 250, 0, 255).astype(np.uint8)),
         ("blobs 255x257", blobs(7, 255, 257, 150)), ("blobs 128x128", blobs(8, 128, 128, 70)), ("random noise 120x160", np.random.default_rng(9).integers(0, 256, (120, 160), dtype=np.uint8))]
sift = SIFT(max_keypoints=16384)
bad = 0
for name, img in cases:
    t0 = time.perf_counter()
    okp, osize, oang, oresp, ooct, odesc = osift.detect_and_compute(img)
    t_or = time.perf_counter() - t0
    try:
        kpt, resp, octv, desc = sift.detect_and_compute_device(img)
        k = kpt.cpu().numpy()
        same = (len(k) == len(okp) and np.array_equal(k[:, :2].view(np.int32), okp.view(np.int32)) and np.array_equal(k[:, 2].view(np.int32), osize.view(np.int32))
                and np.array_equal(k[:, 3].view(np.int32), oang.view(np.int32)) and np.array_equal(resp.cpu().numpy().view(np.int32), oresp.view(np.int32))
                and np.array_equal(octv.cpu().numpy(), ooct) and np.array_equal(desc.cpu().numpy(), odesc))
        print(f"{name}: oracle {len(okp)} keypoints ({t_or:.1f} s), here {len(k)}: {'identical' if same else 'MISMATCH'}", flush=True)
    except Exception as e:  # noqa: BLE001
        same = False
        print(f"{name}: oracle {len(okp)} keypoints, here: {type(e).__name__}: {e}", flush=True)
    bad += not same
print("mismatching cases:", bad)
sys.exit(1 if bad else 0)
