"""Developer tool: wall-clock cost of the cv2-shaped SIFT call a node makes per frame -- `SIFT.detectAndCompute(image, None)` from a numpy image to a
list of KeyPoint objects + descriptors -- next to the device-resident call (`detect_and_compute_device`) and the array-shaped extractor.
   python tools/bench_sift_api.py"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gisnav_amd.sift import SIFT  # noqa: E402
rng = np.random.default_rng(5)
yy, xx = np.mgrid[0:480, 0:640]
img = np.zeros((480, 640))
for _ in range(900):
    cx, cy, s, a = rng.uniform(0, 640), rng.uniform(0, 480), rng.uniform(2, 9), rng.uniform(-80, 80)
    img += a * np.exp(-((xx - cx) ** 2 + (yy - cy) ** 2) / (2 * s * s))
img = np.clip(128 + img + rng.normal(0, 2, img.shape), 0, 255).astype(np.uint8)
sift = SIFT(max_keypoints=8192)
ext = sift.as_extractor()


def timeit(fn, n=100):
    for _ in range(5): r = fn()
    torch.cuda.synchronize(); ts = []
    for _ in range(n):
        t0 = time.perf_counter(); r = fn(); ts.append(time.perf_counter() - t0)
    return np.median(ts) * 1e3, np.percentile(ts, 95) * 1e3, r


a = timeit(lambda: sift.detect_and_compute_device(img))
b = timeit(lambda: ext(img))
c = timeit(lambda: sift.detectAndCompute(img, None))
print(f"480x640 frame, {len(c[2][0])} keypoints: device-resident call median {a[0]:.3f} ms (p95 {a[1]:.3f}); array extractor (numpy out) {b[0]:.3f} ms (p95 {b[1]:.3f}); "
      f"cv2-shaped detectAndCompute (KeyPoint objects) {c[0]:.3f} ms (p95 {c[1]:.3f})")
def as_ref(kps):   # what pose_node.py does with cv2's list next
    return np.array([kp.pt for kp in kps], np.float32), np.array([kp.size for kp in kps], np.float32), np.array([kp.angle for kp in kps], np.float32)
kps = c[2][0]
t0 = time.perf_counter()
for _ in range(50): as_ref(kps)
print(f"   turning that list back into arrays, as the node does next: {(time.perf_counter() - t0) / 50 * 1e3:.3f} ms")
