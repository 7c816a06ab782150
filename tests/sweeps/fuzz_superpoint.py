"""Developer tool: the SuperPoint extractor against its oracle over image sizes the suite does not hold (small, very wide / tall, 8-multiples that are
not 16- or 32-multiples, and sizes that are NOT multiples of 8), both f32-accurate arithmetics: keypoint sets equal up to score ties, scores 1e-5,
descriptors 1e-4.   python tests/sweeps/fuzz_superpoint.py"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import superpoint as osp  # noqa: E402   (checker, as in tests/)
from gisnav_amd.engine import PoseEngine  # noqa: E402
from gisnav_amd.superpoint import SuperPoint  # noqa: E402


def image(seed, h, w):
    rng = np.random.default_rng(seed)
    base = rng.uniform(0, 1, (h // 8 + 3, w // 8 + 3)).astype(np.float32)
    big = torch.nn.functional.interpolate(torch.from_numpy(base)[None, None], size=(h + 16, w + 16), mode="bicubic", align_corners=False)[0, 0].clamp(0, 1)
    return (big[8:8 + h, 8:8 + w] + 0.02 * torch.from_numpy(rng.standard_normal((h, w)).astype(np.float32))).clamp(0, 1).numpy()


sd = osp.synthetic_state_dict(0)
sizes = [(32, 32), (40, 72), (64, 520), (520, 64), (88, 136), (104, 200), (248, 328), (360, 488), (125, 163), (100, 300)]
bad = 0
for prec in ("f32", "f16x2_bf16_attn"):
    eng = PoseEngine(0, max_batch=1, max_kpts=128, precision=prec, feature="superpoint")
    sp = SuperPoint(engine=eng, max_keypoints=1024, state_dict=sd)
    for k, (h, w) in enumerate(sizes):
        img = image(40 + k, h, w)
        try:
            okp, osc, od = osp.detect_and_describe(sd, torch.from_numpy(img), 1024)
            oerr = None
        except Exception as e:  # noqa: BLE001
            oerr = e
        try:
            kpt, score, desc, n = sp.detect_and_describe_device(img[None])
            torch.cuda.synchronize()
            gerr = None
        except Exception as e:  # noqa: BLE001
            gerr = e
        if oerr or gerr:
            both = bool(oerr) and bool(gerr)
            print(f"{prec} {h}x{w}: oracle {'raises ' + type(oerr).__name__ if oerr else 'ok'}; here {'raises ' + str(gerr)[:90] if gerr else 'ok'} {'(both refuse)' if both else ('REFUSED here, reference runs' if gerr else 'MISMATCH')}", flush=True)
            bad += (not both) and not (h % 8 or w % 8)
            continue
        m = int(n[0])
        got = {(float(x), float(y)): i for i, (x, y) in enumerate(kpt[0, :m, :2].cpu().numpy())}
        ref = {(float(x), float(y)): i for i, (x, y) in enumerate(okp.numpy())}
        common = set(got) & set(ref)
        gi = np.array([got[c] for c in common], int); ri = np.array([ref[c] for c in common], int)
        ds = float(np.abs(score[0].cpu().numpy()[gi] - osc.numpy()[ri]).max()) if len(common) else 0.0
        dd = float(np.abs(desc[0].cpu().numpy()[gi] - od.numpy()[ri]).max()) if len(common) else 0.0
        ok = len(common) >= 0.99 * len(ref) and abs(m - len(ref)) <= max(2, len(ref) // 100) and ds < 1e-5 and dd < 1e-4
        bad += not ok
        print(f"{prec} {h}x{w}: oracle {len(ref)} keypoints, here {m}, common {len(common)}, |dscore| {ds:.1e}, |ddesc| {dd:.1e} {'ok' if ok else 'MISMATCH'}", flush=True)
print("mismatching:", bad)
sys.exit(1 if bad else 0)
