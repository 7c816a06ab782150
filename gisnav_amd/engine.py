"""Device-side engine: one `PoseEngine` per GPU wraps a `gn_ctx` of libgisnav_amd.so.

PyTorch is used here only for device memory, streams and (in `dist.py`) the RCCL process
group; every numeric step of the hot path runs in the hand-written gfx950 kernels behind the
C ABI.  The engine mirrors the state PoseNode builds in its constructor
(ros/gisnav/gisnav/core/pose_node.py:81-122): a device, a LightGlue("sift") matcher with 9
layers / threshold 0.5 / no early exit, and the solvePnPRansac settings of
core/_shared.py:109-116.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional, Tuple

import numpy as np
import torch

from . import _lib
from .weights import canonical_state_dict

MIN_MATCHES = 15          # PoseNode.MIN_MATCHES, pose_node.py:63
FILTER_THRESHOLD = 0.5    # PoseNode.CONFIDENCE_THRESHOLD, pose_node.py:60
RANSAC_ITERATIONS = 10    # _shared.py:115
RANSAC_REPROJ_PX = 8.0    # cv2.solvePnPRansac default
RANSAC_CONFIDENCE = 0.99  # cv2.solvePnPRansac default

_PRECISIONS = {"f32": _lib.GN_PREC_F32, "bf16_attn": _lib.GN_PREC_BF16_ATTN, "f32x3_bf16_attn": _lib.GN_PREC_F32X3_BF16_ATTN,
               "f16x2_bf16_attn": _lib.GN_PREC_F16X2_BF16_ATTN, "f16x2_f16_attn": _lib.GN_PREC_F16X2_F16_ATTN}


def _ptr(t: Optional[torch.Tensor]) -> C.c_void_p:
    return C.c_void_p(0 if t is None else t.data_ptr())


def _dev_tensor(t, dtype, device) -> torch.Tensor:
    if isinstance(t, np.ndarray):
        t = torch.from_numpy(np.ascontiguousarray(t))
    t = t.to(device=device, dtype=dtype, non_blocking=True)
    return t.contiguous()


def numa_cpus(node: int):
    """CPU ids of NUMA node `node` (/sys/devices/system/node/node<N>/cpulist, e.g. "0-31,128-159"), or None."""
    try:
        with open(f"/sys/devices/system/node/node{int(node)}/cpulist") as f:
            return parse_cpulist(f.read())
    except (OSError, ValueError):
        return None


def parse_cpulist(text: str):
    cpus = set()
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        cpus.update(range(int(lo), int(hi or lo) + 1))
    return cpus or None


class RecordStager:
    """Input side of a STREAM of new frame<->tile pairs (pose_node.py:207-213, 254-265): the raw `query_sift` / reference keypoint payloads
    (532-byte KEYPOINT_DTYPE records) and the DEM rasters of a batch are copied -- bytes, no unpacking -- into pinned host buffers and
    uploaded on a copy stream into one of `depth` device slots, while the engine works on the previous batch; `k_prep` reads the records
    as they are (GN_KPT_RECORD).  The reference builds per message: np.frombuffer + np.column_stack + six torch.tensor(...).to(device)
    calls from pageable memory.

        nxt = stager.stage(msgs)          # host memcpy into pinned memory + async H2D (any thread)
        stager.wait(cur); eng.estimate(cur, K, out=out); stager.release(cur)

    Contract: every staged batch is release()d on the compute stream after the last kernel that reads it and BEFORE the stage() call that
    re-uses its slot is issued (with `depth` slots: before the depth-th following stage()); a slot whose batch was never released is
    overwritten without waiting.
    """

    def __init__(self, engine: "PoseEngine", max_batch: int, max_kpts: int, dem_hw: Tuple[int, int], depth: int = 3, copy_threads: int = 4,
                 numa: Optional[str] = "auto"):
        import os
        from concurrent.futures import ThreadPoolExecutor
        # one rank per GPU: the copy threads of a rank stay on the cores of its GPU's NUMA node (host staging is ~6 ms of memcpy
        # per 5.8 ms step and rank; eight unpinned pools wander over both sockets).  "auto": gn_device_numa_node; None: leave the threads alone.
        self.numa_node, cpus = -1, None
        if numa == "auto":
            self.numa_node = int(engine.lib.gn_device_numa_node(engine.device.index or 0))
            cpus = numa_cpus(self.numa_node) if self.numa_node >= 0 else None
            if cpus:
                cpus = (cpus & os.sched_getaffinity(0)) or None      # never outside what the launcher allowed this process

        def _pin():
            if cpus:
                try:
                    os.sched_setaffinity(0, cpus)
                except OSError:
                    pass
        self._pin = _pin
        self.eng, self.B, self.K, self.depth = engine, int(max_batch), int(max_kpts), int(depth)
        # the pinned-memory copies of a batch (35 MB at 32 x 1024 keypoints per side) run on a few host threads: one thread's memcpy
        # (~6 GB/s) would be slower than the GPU's step; numpy releases the GIL inside the copies
        self._pool = ThreadPoolExecutor(max_workers=max(1, int(copy_threads)), initializer=_pin) if copy_threads > 1 else None
        dev, (H, W) = engine.device, dem_hw
        self.copy_stream = torch.cuda.Stream(device=dev)
        mk = lambda shape, dt, **kw: torch.empty(shape, dtype=dt, **kw)  # noqa: E731
        self.slots = []
        for _ in range(self.depth):
            host = dict(rec_q=mk((self.B, self.K, 133), torch.float32, pin_memory=True), rec_r=mk((self.B, self.K, 133), torch.float32, pin_memory=True),
                        dem=mk((self.B, H, W), torch.uint8, pin_memory=True), n=mk((2, self.B), torch.int32, pin_memory=True))
            devb = {k: torch.empty_like(v, device=dev) for k, v in host.items()}
            self.slots.append(dict(host=host, host_np={k: v.numpy() for k, v in host.items()}, dev=devb,
                                   uploaded=torch.cuda.Event(), consumed=torch.cuda.Event(), used=False))
        self._next = 0

    def stage(self, msgs) -> dict:
        """msgs: up to max_batch tuples (query_sift bytes, reference-keypoint bytes, dem (H, W) uint8).  Returns the inputs dict of
        `PoseEngine.estimate`; call wait() on the compute stream before using it and release() after the last kernel that reads it."""
        slot = self.slots[self._next]
        h = slot["host_np"]
        B = len(msgs)
        if B < 1 or B > self.B:
            raise _lib.GnError(f"{B} pairs staged, the stager holds 1..{self.B}")
        for b, (q, r, dem) in enumerate(msgs):      # validated BEFORE anything is copied: a refused batch leaves the slot as it was
            if len(q) % 532 or len(r) % 532:
                raise _lib.GnError(f"pair {b}: keypoint payloads must be whole 532-byte KEYPOINT_DTYPE records (got {len(q)} and {len(r)} bytes)")
            if max(len(q), len(r)) // 532 > self.K:
                raise _lib.GnError(f"pair {b}: {max(len(q), len(r)) // 532} keypoints exceed the stager's max_kpts {self.K}")
            if tuple(np.shape(dem)) != tuple(h["dem"].shape[1:]):
                raise _lib.GnError(f"pair {b}: DEM raster {tuple(np.shape(dem))}, the stager was built for {tuple(h['dem'].shape[1:])}")

        self._next = (self._next + 1) % self.depth
        if slot["used"]:
            slot["uploaded"].synchronize()           # the pinned host buffer is free again: its H2D copy has finished
            if slot.get("released", False):
                slot["consumed"].synchronize()       # the batch that last used this slot has been read by its kernels
            else:
                torch.cuda.synchronize(self.eng.device)   # release() was never called for it: wait for everything rather than overwrite a batch in use
        slot["released"] = False

        def copy_pair(b):
            q, r, dem = msgs[b]
            nq, nr = len(q) // 532, len(r) // 532
            h["rec_q"][b, :nq] = np.frombuffer(q, dtype=np.float32).reshape(nq, 133)      # one memcpy: the wire bytes ARE the device layout
            h["rec_r"][b, :nr] = np.frombuffer(r, dtype=np.float32).reshape(nr, 133)
            h["dem"][b] = dem
            h["n"][0, b], h["n"][1, b] = nq, nr

        if self._pool is not None and B > 1:
            list(self._pool.map(copy_pair, range(B)))
        else:
            for b in range(B):
                copy_pair(b)
        d = slot["dev"]
        with torch.cuda.stream(self.copy_stream):
            for k in ("rec_q", "rec_r", "dem", "n"):
                d[k].copy_(slot["host"][k], non_blocking=True)
            slot["uploaded"].record(self.copy_stream)
        slot["used"] = True
        return dict(kpt_q=d["rec_q"][:B], kpt_r=d["rec_r"][:B], n_q=d["n"][0, :B], n_r=d["n"][1, :B], dem=d["dem"][:B],
                    desc_q=None, desc_r=None, kpt_format=_lib.GN_KPT_RECORD, _slot=slot)

    def wait(self, inputs: dict) -> None:
        torch.cuda.current_stream(self.eng.device).wait_event(inputs["_slot"]["uploaded"])

    def release(self, inputs: dict) -> None:
        inputs["_slot"]["consumed"].record(torch.cuda.current_stream(self.eng.device))
        inputs["_slot"]["released"] = True

    def close(self) -> None:
        """Stop the copy threads (the pinned buffers and device slots go with the object)."""
        if self._pool is not None:
            self._pool.shutdown(wait=True)
            self._pool = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001  (interpreter shutdown)
            pass


class PoseEngine:
    """Batched frame<->tile matcher + PnP solver on one MI355X."""

    def __init__(self, device: int = 0, max_batch: int = 32, max_kpts: int = 1024,
                 precision: str = "f32", state_dict: Optional[Dict[str, np.ndarray]] = None,
                 n_layers: int = 9, filter_threshold: float = FILTER_THRESHOLD, guard: str = "flag", feature: str = "sift"):
        if not torch.cuda.is_available():
            raise _lib.GnError("PoseEngine needs a HIP device; the product path has no CPU fallback")
        self.lib = _lib.load()
        self.device = torch.device("cuda", device)
        self.max_batch, self.precision = max_batch, precision
        self._n_layers, self._filter_threshold, self._state_dict = n_layers, filter_threshold, None
        self._guard = {"off": 0, "flag": 1, "sync": 2}[guard]
        self.feature = feature
        self._feature = {"sift": _lib.GN_FEATURE_SIFT, "superpoint": _lib.GN_FEATURE_SUPERPOINT}[feature]
        self.desc_dim = 128 if feature == "sift" else 256
        ctx = C.c_void_p()
        _lib.check(None, self.lib.gn_create_ex(device, max_batch, max_kpts, _PRECISIONS[precision], self._feature, C.byref(ctx)), "gn_create_ex")
        self.ctx = ctx
        self.kmax = self.lib.gn_kmax(ctx)
        _lib.check(ctx, self.lib.gn_set_num_layers(ctx, n_layers), "gn_set_num_layers")
        _lib.check(ctx, self.lib.gn_set_filter_threshold(ctx, filter_threshold), "gn_set_filter_threshold")
        _lib.check(ctx, self.lib.gn_set_guard(ctx, self._guard), "gn_set_guard")
        if state_dict is not None:
            self.load_state_dict(state_dict)

    def guard_status(self):
        """(last matcher call left the fp16 range of the f16x2 mode, number of such calls so far) -- synchronises the stream."""
        last, total = C.c_int32(0), C.c_int64(0)
        _lib.check(self.ctx, self.lib.gn_get_guard_status(self.ctx, self._stream(), C.byref(last), C.byref(total)), "gn_get_guard_status")
        return bool(last.value), int(total.value)

    # ------------------------------------------------------------------ margin certificate (gn_set_certify)
    def set_certify(self, mode, eps: Optional[float] = None, eps_f32: Optional[float] = None) -> None:
        """"off" | "flag" | "rerun" (0 / 1 / 2).  "rerun": every match() / estimate() call synchronises once, and the pairs in which a decision
        lies within `eps` of flipping (or whose activations left the fp16 range) are run again on the exact-f32 kernels before the call
        returns -- correspondence indices are then the exact arithmetic's (pose_node.py:285-297 consumes them as exact integers).
        "deferred" (3; estimate() with sub-batch streams): the flags of call n are read -- and its flagged pairs re-run -- after call n + 1 has been
        enqueued, so the host never waits for an idle GPU.  Contract: the inputs AND outputs of call n stay alive and untouched until call n + 1 has
        returned or flush() was called (alternate two output dicts); results of call n are final from then on."""
        m = {"off": 0, "flag": 1, "rerun": 2, "deferred": 3}.get(mode, mode)
        self._certify_mode, self._last_out_ptr = int(m), None
        _lib.check(self.ctx, self.lib.gn_set_certify(self.ctx, int(m), -1.0 if eps is None else float(eps), -1.0 if eps_f32 is None else float(eps_f32)),
                   "gn_set_certify")

    def fused_projection_status(self) -> int:
        """gn_fused_projection_status: 1 the projection fused behind the block tail was checked bitwise equal to the separate launches on this context's
        weights, 0 it differed and was switched off, -1 not run yet / not applicable."""
        return int(self.lib.gn_fused_projection_status(self.ctx))

    def set_ffn_products(self, products) -> None:
        """gn_set_ffn_products: 3 (default, f32-accurate), 2 (activations' fp16 high term only; run it under set_certify), or "auto" (0): the level
        follows the certificate -- calibrate_certify then measures eps for both levels, and under set_certify("rerun" / "deferred") the context runs on
        two products only while that flags (at most 1 pair in 64) no more pairs than three products would.  Indices do not depend on the level."""
        p = 0 if products in ("auto", 0) else int(products)
        self._ffn_auto = p == 0
        _lib.check(self.ctx, self.lib.gn_set_ffn_products(self.ctx, p), "gn_set_ffn_products")

    def set_ffn_level_eps(self, eps_two: float, eps_three: float, level: int = 3) -> None:
        """gn_set_ffn_level_eps: both levels' eps as an earlier calibrate_certify (same weights, same batch size) measured them, and the level to start on."""
        _lib.check(self.ctx, self.lib.gn_set_ffn_level_eps(self.ctx, float(eps_two), float(eps_three), int(level)), "gn_set_ffn_level_eps")

    def ffn_level(self) -> Dict[str, float]:
        """gn_get_ffn_level: the level the next call runs on, the calibrated eps of both levels, and how many certified calls ran on each."""
        lvl, e2, e3, buf = C.c_int32(0), C.c_float(0.0), C.c_float(0.0), (C.c_int64 * 4)()
        _lib.check(self.ctx, self.lib.gn_get_ffn_level(self.ctx, C.byref(lvl), C.byref(e2), C.byref(e3), buf), "gn_get_ffn_level")
        return {"level": int(lvl.value), "eps_two_products": float(e2.value), "eps_three_products": float(e3.value),
                "calls_two_products": int(buf[0]), "calls_three_products": int(buf[1]), "switches": int(buf[2]), "automatic": bool(buf[3])}

    def calibrate_certify(self, inputs: dict, safety: float = 4.0, floor_eps: float = 1.0e-5) -> Dict[str, float]:
        """gn_calibrate_certify on staged inputs (the dict of stage_inputs / RecordStager.stage): measures max |P_mode - P_f32| over the deciding
        entries of the sample and sets eps = max(floor_eps, safety * that).  Returns {"measured": ..., "eps": ...}."""
        B = inputs["kpt_q"].shape[0]
        m, e = C.c_float(0.0), C.c_float(0.0)
        rc = self.lib.gn_calibrate_certify(self.ctx, B, inputs["kpt_format"],
                                           _ptr(inputs.get("desc_q")), _ptr(inputs["kpt_q"]), _ptr(inputs["n_q"]), inputs["kpt_q"].shape[1],
                                           _ptr(inputs.get("desc_r")), _ptr(inputs["kpt_r"]), _ptr(inputs["n_r"]), inputs["kpt_r"].shape[1],
                                           float(safety), float(floor_eps), C.byref(m), C.byref(e), self._stream())
        _lib.check(self.ctx, rc, "gn_calibrate_certify")
        lv = self.ffn_level()
        d = {"measured": float(m.value), "eps": float(e.value), "safety": float(safety)}
        if getattr(self, "_ffn_auto", False):      # set_ffn_products("auto"): both levels were measured; measured / eps are the three-product level's
            d["eps_two_products"], d["eps_three_products"] = lv["eps_two_products"], lv["eps_three_products"]
        return d

    def certify_stats(self, reset: bool = False) -> Dict[str, int]:
        buf = (C.c_int64 * 8)()
        _lib.check(self.ctx, self.lib.gn_get_certify_stats(self.ctx, buf), "gn_get_certify_stats")
        if reset:
            _lib.check(self.ctx, self.lib.gn_reset_certify_stats(self.ctx), "gn_reset_certify_stats")
        keys = ("calls", "pairs", "flagged_margin", "flagged_fp16_range", "rerun_pairs", "f32_marginal_pairs", "mode")
        d = {k: int(buf[i]) for i, k in enumerate(keys)}
        d["rerun_fraction"] = (d["rerun_pairs"] / d["pairs"]) if d["pairs"] else 0.0
        return d

    def uncertain(self, B: int) -> np.ndarray:
        """Per-pair flags of the most recent matcher call (0 certified, 1 margin, 2 fp16 range); synchronises the stream."""
        out = np.zeros(B, np.int32)
        _lib.check(self.ctx, self.lib.gn_get_uncertain(self.ctx, int(B), out.ctypes.data_as(_lib.c_i32p), self._stream()), "gn_get_uncertain")
        return out

    def __del__(self):
        ctx, self.ctx = getattr(self, "ctx", None), None
        if ctx:
            self.lib.gn_destroy(ctx)

    # ------------------------------------------------------------------ weights
    def grow(self, max_kpts: int) -> None:
        """Re-size the context for more keypoints per side (`gn_resize`: the workspaces are replaced, every weight, SuperPoint tensor and
        setting stays).  The reference accepts any keypoint count (cv2.SIFT_create() is unbounded, pose_node.py:122); the mirrors call this
        instead of failing on a larger cloud.  Costs a device synchronisation and a few allocations -- no weight reload (round 2 re-created
        the whole context: a multi-100 ms stall on the first large tile)."""
        if max_kpts <= self.kmax:
            return
        _lib.check(self.ctx, self.lib.gn_resize(self.ctx, int(max_kpts)), "gn_resize")
        self.kmax = self.lib.gn_kmax(self.ctx)
        self._sift = None

    # SuperPoint weights / arithmetic live in the context (gn_sp_*) and survive gn_resize
    def sp_set_arithmetic(self, mode: int) -> None:
        _lib.check(self.ctx, self.lib.gn_sp_set_arithmetic(self.ctx, int(mode)), "gn_sp_set_arithmetic")

    def sp_load_state_dict(self, sd) -> None:
        for name, arr in sd.items():
            if hasattr(arr, "detach"):
                arr = arr.detach().cpu().numpy()
            arr = np.ascontiguousarray(arr, dtype=np.float32)
            shape = (C.c_int64 * max(arr.ndim, 1))(*(arr.shape if arr.ndim else (1,)))
            rc = self.lib.gn_sp_load_tensor(self.ctx, name.encode(), arr.ctypes.data_as(C.c_void_p), shape, max(arr.ndim, 1))
            _lib.check(self.ctx, rc, f"gn_sp_load_tensor({name})")

    def load_state_dict(self, sd) -> None:
        self._state_dict = sd
        for name, arr in canonical_state_dict(sd).items():
            arr = np.ascontiguousarray(arr, dtype=np.float32)
            shape = (C.c_int64 * max(arr.ndim, 1))(*(arr.shape if arr.ndim else (1,)))
            rc = self.lib.gn_load_tensor(self.ctx, name.encode(), arr.ctypes.data_as(C.c_void_p), shape, max(arr.ndim, 1))
            _lib.check(self.ctx, rc, f"gn_load_tensor({name})")
        missing = self.lib.gn_missing_tensors(self.ctx)
        if missing:
            raise _lib.GnError(f"{missing} required LightGlue tensors missing from the state dict")

    def set_image_size(self, wh_q=None, wh_r=None) -> None:
        """kornia LightGlueMatcher's hw1 / hw2 as (w, h): image sizes for the keypoint normalisation; None = keypoint extent (what
        PoseNode gets, pose_node.py:285-287)."""
        q, r = wh_q or (0.0, 0.0), wh_r or (0.0, 0.0)
        _lib.check(self.ctx, self.lib.gn_set_image_size(self.ctx, float(q[0]), float(q[1]), float(r[0]), float(r[1])), "gn_set_image_size")

    def set_num_layers(self, n: int) -> None:
        self._n_layers = n
        _lib.check(self.ctx, self.lib.gn_set_num_layers(self.ctx, n), "gn_set_num_layers")

    # ------------------------------------------------------------------ helpers
    def _stream(self) -> C.c_void_p:
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def stage_inputs(self, pairs, kpt_format: int = _lib.GN_KPT_XYSA):
        """Pack a list of `synthetic.Pair`-like objects into device tensors (host -> HBM once)."""
        B = len(pairs)
        nq = max(len(p.kp_q) for p in pairs)
        nr = max(len(p.kp_r) for p in pairs)
        D = self.desc_dim
        dq = np.zeros((B, nq, D), np.float32); kq = np.zeros((B, nq, 4), np.float32)
        dr = np.zeros((B, nr, D), np.float32); kr = np.zeros((B, nr, 4), np.float32)
        n_q = np.zeros(B, np.int32); n_r = np.zeros(B, np.int32)
        h, w = pairs[0].dem.shape
        dem = np.zeros((B, h, w), np.uint8)
        for b, p in enumerate(pairs):
            a, c = len(p.kp_q), len(p.kp_r)
            n_q[b], n_r[b] = a, c
            dq[b, :a], dr[b, :c] = p.desc_q, p.desc_r
            kq[b, :a] = np.column_stack([p.kp_q, p.size_q, p.angle_q])
            kr[b, :c] = np.column_stack([p.kp_r, p.size_r, p.angle_r])
            dem[b] = p.dem
        f = lambda a, dt: _dev_tensor(a, dt, self.device)  # noqa: E731
        return dict(desc_q=f(dq, torch.float32), kpt_q=f(kq, torch.float32), n_q=f(n_q, torch.int32),
                    desc_r=f(dr, torch.float32), kpt_r=f(kr, torch.float32), n_r=f(n_r, torch.int32),
                    dem=f(dem, torch.uint8), kpt_format=kpt_format)

    # ------------------------------------------------------------------ hot path
    def match(self, desc_q, kpt_q, n_q, desc_r, kpt_r, n_r, kpt_format: int = _lib.GN_KPT_XYSA,
              out: Optional[tuple] = None):
        """gn_match.  Tensors are device tensors: desc [B,S,128] f32, kpt [B,S,4|6] f32, n [B] i32.
        Returns (idx [B,kmax,2] i64, score [B,kmax] f32, n_match [B] i32) on the device."""
        B = kpt_q.shape[0]
        if out is None:
            idx = torch.empty((B, self.kmax, 2), dtype=torch.int64, device=self.device)
            score = torch.empty((B, self.kmax), dtype=torch.float32, device=self.device)
            n_match = torch.empty((B,), dtype=torch.int32, device=self.device)
        else:
            idx, score, n_match = out
        rc = self.lib.gn_match(self.ctx, B, kpt_format, _ptr(desc_q), _ptr(kpt_q), _ptr(n_q), kpt_q.shape[1],
                               _ptr(desc_r), _ptr(kpt_r), _ptr(n_r), kpt_r.shape[1],
                               _ptr(idx), _ptr(score), _ptr(n_match), self._stream())
        _lib.check(self.ctx, rc, "gn_match")
        return idx, score, n_match

    def gather_points(self, kpt_q, kpt_r, idx, n_match, dem, kpt_format: int = _lib.GN_KPT_XYSA):
        B = kpt_q.shape[0]
        mkp_q = torch.zeros((B, self.kmax, 2), dtype=torch.float32, device=self.device)
        obj = torch.zeros((B, self.kmax, 3), dtype=torch.float32, device=self.device)
        H, W = (dem.shape[1], dem.shape[2]) if dem is not None else (0, 0)
        rc = self.lib.gn_gather_points(self.ctx, B, kpt_format, _ptr(kpt_q), kpt_q.shape[1], _ptr(kpt_r), kpt_r.shape[1],
                                       _ptr(idx), _ptr(n_match), _ptr(dem), H, W, _ptr(mkp_q), _ptr(obj), self._stream())
        _lib.check(self.ctx, rc, "gn_gather_points")
        return mkp_q, obj

    def pnp_ransac(self, obj, img, n_pts, K: np.ndarray, iterations: int = RANSAC_ITERATIONS,
                   reproj_px: float = RANSAC_REPROJ_PX, confidence: float = RANSAC_CONFIDENCE, min_pts: int = 5):
        """gn_pnp_ransac.  obj [B,S,3] f32, img [B,S,2] f32, n_pts [B] i32 (device)."""
        B = obj.shape[0]
        R = torch.empty((B, 3, 3), dtype=torch.float64, device=self.device)
        t = torch.empty((B, 3, 1), dtype=torch.float64, device=self.device)
        n_inl = torch.empty((B,), dtype=torch.int32, device=self.device)
        ok = torch.empty((B,), dtype=torch.uint8, device=self.device)
        K9 = np.ascontiguousarray(np.asarray(K, np.float64).reshape(9))
        rc = self.lib.gn_pnp_ransac(self.ctx, B, _ptr(obj), _ptr(img), _ptr(n_pts), obj.shape[1],
                                    K9.ctypes.data_as(_lib.c_f64p), iterations, reproj_px, confidence, min_pts,
                                    _ptr(R), _ptr(t), _ptr(n_inl), _ptr(ok), self._stream())
        _lib.check(self.ctx, rc, "gn_pnp_ransac")
        return R, t, n_inl, ok

    def to_device(self, name: str, array, dtype: torch.dtype = torch.float32) -> torch.Tensor:
        """Host array -> device tensor through this engine's pinned staging (gisnav_amd.upload.PinnedUploader); a device tensor passes through."""
        if isinstance(array, torch.Tensor):
            return array
        if getattr(self, "_uploader", None) is None:
            from .upload import PinnedUploader
            self._uploader = PinnedUploader(self.device)
        return self._uploader(name, array, dtype)

    def to_host(self, *tensors: torch.Tensor):
        """Small device tensors -> numpy copies through one pinned block and one synchronisation (gisnav_amd.upload.PinnedDownloader)."""
        if getattr(self, "_downloader", None) is None:
            from .upload import PinnedDownloader
            self._downloader = PinnedDownloader(self.device)
        return self._downloader(*tensors)

    def pnp_ransac_host(self, obj: np.ndarray, img: np.ndarray, K: np.ndarray, iterations: int = RANSAC_ITERATIONS,
                        reproj_px: float = RANSAC_REPROJ_PX, confidence: float = RANSAC_CONFIDENCE, min_pts: int = 5):
        """gn_pnp_ransac for ONE correspondence list given as host arrays (seam B2): obj (n, 3), img (n, 2) -> (R (3,3) f64, t (3,1) f64, n_inliers, ok)
        as host values.  One pinned staging block in ([n | pad | obj | img], one asynchronous copy), one 112-byte block out: pageable transfers of
        this size were measured to stall for ~90 ms every few dozen calls on the MI355X boxes (tools/bench_seams.py), pinned ones never."""
        n = int(len(obj))
        if n > self.kmax:
            raise _lib.GnError(f"{n} correspondences exceed this context's max_kpts {self.kmax}")
        io = getattr(self, "_pnp_io", None)
        if io is None or io["cap"] != self.kmax:
            cap = self.kmax
            pin = torch.empty(8 + 5 * cap, dtype=torch.float32, pin_memory=True)
            out = torch.zeros(112, dtype=torch.uint8, device=self.device)
            io = self._pnp_io = dict(cap=cap, pin=pin, pin_np=pin.numpy(), dev=torch.empty(8 + 5 * cap, dtype=torch.float32, device=self.device), out=out,
                                     host=torch.empty(112, dtype=torch.uint8, pin_memory=True))
            io["host_np"] = io["host"].numpy()
        h = io["pin_np"]
        h[:1].view(np.int32)[0] = n
        i0 = 4 + (3 * n + 3) // 4 * 4                       # the image points start on a 16-byte boundary, like the object points
        h[4:4 + 3 * n] = np.asarray(obj, np.float32).reshape(-1)
        h[i0:i0 + 2 * n] = np.asarray(img, np.float32).reshape(-1)
        d, o = io["dev"], io["out"]
        d[:i0 + 2 * n].copy_(io["pin"][:i0 + 2 * n], non_blocking=True)
        K9 = np.ascontiguousarray(np.asarray(K, np.float64).reshape(9))
        rc = self.lib.gn_pnp_ransac(self.ctx, 1, _ptr(d[4:4 + 3 * n]), _ptr(d[i0:i0 + 2 * n]), _ptr(d[:1].view(torch.int32)), n,
                                    K9.ctypes.data_as(_lib.c_f64p), iterations, reproj_px, confidence, min_pts,
                                    _ptr(o[0:72]), _ptr(o[72:96]), _ptr(o[100:104]), _ptr(o[104:105]), self._stream())
        _lib.check(self.ctx, rc, "gn_pnp_ransac")
        io["host"].copy_(o, non_blocking=True)
        torch.cuda.current_stream(self.device).synchronize()
        r = io["host_np"]
        return (r[0:72].view(np.float64).reshape(3, 3).copy(), r[72:96].view(np.float64).reshape(3, 1).copy(), int(r[100:104].view(np.int32)[0]), bool(r[104]))

    def set_ragged(self, ragged: Optional[bool]) -> None:
        """Override of a choice the library makes by itself.  The big kernels skip tiles that hold only padding either way; the block tail has two
        forms (one workgroup per tile, or one per CU walking the list of non-empty tiles: ~20 % faster on batches whose keypoint counts differ a lot --
        an unbounded cv2.SIFT_create(), pose_node.py:122 --, 3 % slower per launch on batches without padding; identical bits).  By default
        (`None`) the library picks the form per call from the fraction of padding-only tiles its previous call saw; True / False force one."""
        _lib.check(self.ctx, self.lib.gn_debug_set_variant(self.ctx, 31, 1 if ragged is None else 2 if ragged else 3), "gn_debug_set_variant(31)")

    def set_overlap(self, enable: bool) -> None:
        """Batch-serving option: PnP of call n overlaps the matcher of call n+1 (see gn_set_overlap); call flush()
        before reading R / t / n_inliers / ok."""
        _lib.check(self.ctx, self.lib.gn_set_overlap(self.ctx, int(enable)), "gn_set_overlap")

    def set_substreams(self, n: int, deferred_join: bool = False) -> None:
        """Throughput option: every estimate() call runs its pairs as n groups on internal streams (gn_set_substreams).  With
        deferred_join the groups are only joined by flush(): keep the input tensors of a call alive until then."""
        _lib.check(self.ctx, self.lib.gn_set_substreams(self.ctx, int(n)), "gn_set_substreams")
        _lib.check(self.ctx, self.lib.gn_set_deferred_join(self.ctx, int(bool(deferred_join))), "gn_set_deferred_join")

    def set_active_kpts(self, max_kpts_per_side: int) -> int:
        """Padded keypoint count the following match()/estimate() calls run at (gn_set_active_kpts): pass the largest keypoint
        count of the batch when it is well below max_kpts.  Returns the padded size in effect."""
        rc = self.lib.gn_set_active_kpts(self.ctx, int(max_kpts_per_side))
        if rc < 0:
            _lib.check(self.ctx, rc, "gn_set_active_kpts")
        return rc

    def flush(self) -> None:
        _lib.check(self.ctx, self.lib.gn_flush(self.ctx, self._stream()), "gn_flush")
        self._last_out_ptr = None

    def estimate(self, inputs: dict, K: np.ndarray, min_matches: int = MIN_MATCHES, out: Optional[dict] = None):
        """gn_estimate on staged inputs: PoseNode._pose lines 246-308 for the whole batch."""
        B = inputs["kpt_q"].shape[0]
        if out is None:
            out = self.alloc_outputs(B)
        if getattr(self, "_certify_mode", 0) == 3:
            if self._last_out_ptr == out["R"].data_ptr():
                raise _lib.GnError("deferred certificate: this call would write the outputs of the previous call, whose flagged pairs may still be re-run -- "
                                   "alternate two output dicts, or flush() between the calls")
            self._last_out_ptr = out["R"].data_ptr()
        dem = inputs.get("dem")
        H, W = (dem.shape[1], dem.shape[2]) if dem is not None else (0, 0)
        K9 = np.ascontiguousarray(np.asarray(K, np.float64).reshape(9))
        # (GN_KPT_RECORD inputs carry the descriptors inside the keypoint records: desc_q / desc_r are None)
        rc = self.lib.gn_estimate(self.ctx, B, inputs["kpt_format"],
                                  _ptr(inputs.get("desc_q")), _ptr(inputs["kpt_q"]), _ptr(inputs["n_q"]), inputs["kpt_q"].shape[1],
                                  _ptr(inputs.get("desc_r")), _ptr(inputs["kpt_r"]), _ptr(inputs["n_r"]), inputs["kpt_r"].shape[1],
                                  _ptr(dem), H, W, K9.ctypes.data_as(_lib.c_f64p), min_matches,
                                  _ptr(out["R"]), _ptr(out["t"]), _ptr(out["n_match"]), _ptr(out["n_inliers"]), _ptr(out["ok"]),
                                  self._stream())
        _lib.check(self.ctx, rc, "gn_estimate")
        return out

    def estimate_bucketed(self, inputs: dict, K: np.ndarray, n_q_host, n_r_host, bucket_pairs: int = 8, min_matches: int = MIN_MATCHES,
                          out: Optional[dict] = None):
        """Length-bucketing scheduler for ragged batches: `cv2.SIFT_create()` is unbounded (pose_node.py:122), so the keypoint counts of a
        batch of messages spread widely, and one gn_estimate call pads every pair to the batch maximum (attention cost grows with the
        SQUARE of the padded length).  The pairs are sorted by max(n_q, n_r) (host-side counts: no device read), run as groups of
        `bucket_pairs` consecutive pairs each padded to ITS maximum (gn_set_active_kpts), and the results are returned in the caller's
        order.  Results do not depend on the padding (tests).  They DO depend, at rounding level, on the number of pairs per call: calls of one or
        two pairs (and sub-stream groups of that size) run the small-grid kernel family, whose summation order differs from the bulk kernels' -- a
        remainder bucket of one or two pairs therefore gives the same matches and poses up to rounding, not the same bits, as one padded call
        (tests/test_gpu_round6.py).  With the certificate on (set_certify) the correspondence indices are the exact arithmetic's either
        way.  Returns (out, stats)."""
        n_q_host = np.asarray(n_q_host).astype(np.int64)
        n_r_host = np.asarray(n_r_host).astype(np.int64)
        B = len(n_q_host)
        if out is None:
            out = self.alloc_outputs(B)
        need = np.maximum(np.maximum(n_q_host, n_r_host), 1)
        order = np.argsort(-need, kind="stable")
        perm = torch.as_tensor(order, device=self.device)
        srt = {k: (v.index_select(0, perm) if isinstance(v, torch.Tensor) else v) for k, v in inputs.items()}
        tmp = self.alloc_outputs(B)
        pad_tokens = 0
        try:
            for a in range(0, B, bucket_pairs):
                b = min(B, a + bucket_pairs)
                pad = self.set_active_kpts(int(need[order[a:b]].max()))
                pad_tokens += 2 * pad * (b - a)
                grp = {k: (v[a:b] if isinstance(v, torch.Tensor) else v) for k, v in srt.items()}
                self.estimate(grp, K, min_matches, out={k: v[a:b] for k, v in tmp.items()})
        finally:
            self.set_active_kpts(self.kmax)
        # with set_overlap / set_substreams(deferred_join=True) the PnP stage or the groups may still run on the library's internal streams:
        # join them before `tmp` is read on the caller's stream and before the index_select'ed copies in `srt` are released
        self.flush()
        for k in out:
            out[k].index_copy_(0, perm, tmp[k])
        real = int((n_q_host + n_r_host).sum())
        one_call_pad = 2 * B * (((int(need.max()) + 127) // 128) * 128)
        return out, {"real_tokens": real, "padded_tokens_bucketed": int(pad_tokens), "padded_tokens_one_call": int(one_call_pad),
                     "groups": (B + bucket_pairs - 1) // bucket_pairs}

    def estimate_images(self, frames, tiles, K: np.ndarray, dem=None, sift=None, min_matches: int = MIN_MATCHES, out: Optional[dict] = None):
        """Frames -> pose from pixels for B pairs, everything in HBM: one batched SIFT pass over the B camera frames and the
        B map tiles (`gn_sift_detect_and_compute_batch`), the matcher padded to what the batch needs (`gn_set_active_kpts`),
        then `gn_estimate`.  frames / tiles: (B, H, W) uint8 (numpy or device tensors, one size); dem: (B, H, W) uint8 or
        None (flat).  Returns (estimate outputs, n_keypoints [2B] int32 on the host: frames first)."""
        from .sift import SIFT
        if sift is None:
            if getattr(self, "_sift", None) is None:
                self._sift = SIFT(engine=self, max_keypoints=self.kmax)
            sift = self._sift
        to_dev = lambda a: a if isinstance(a, torch.Tensor) else torch.as_tensor(np.ascontiguousarray(a, np.uint8), device=self.device)  # noqa: E731
        f, t = to_dev(frames), to_dev(tiles)
        assert f.shape == t.shape and f.dim() == 3, "expected two (B, H, W) uint8 stacks of one size"
        B = int(f.shape[0])
        kpt, _, _, desc, n = sift.detect_and_compute_batch_device(torch.cat([f, t], 0))
        totals = sift.last_totals(2 * B)
        if int(totals.max()) > sift._max:
            # cv2.SIFT_create() is unbounded (pose_node.py:122); the extractor kept the strongest max_keypoints by response.  Say so
            # instead of silently matching a truncated cloud; a caller that wants them all passes a larger SIFT / engine.
            import warnings
            warnings.warn(f"SIFT found up to {int(totals.max())} keypoints per image, kept the {sift._max} strongest (max_keypoints); "
                          "create the PoseEngine / SIFT with a larger max_kpts to match them all", RuntimeWarning, stacklevel=2)
        nd = torch.as_tensor(n, device=self.device)
        if dem is None:
            dem = torch.zeros((B, int(f.shape[1]), int(f.shape[2])), dtype=torch.uint8, device=self.device)
        inputs = dict(desc_q=desc[:B], kpt_q=kpt[:B], n_q=nd[:B], desc_r=desc[B:], kpt_r=kpt[B:], n_r=nd[B:], dem=to_dev(dem), kpt_format=_lib.GN_KPT_XYSA)
        self.set_active_kpts(max(int(n.max()), 1))
        try:
            res = self.estimate(inputs, K, min_matches, out=out)
            self.flush()                                       # sub-stream groups read the temporaries above: join before they are released
        finally:
            self.set_active_kpts(self.kmax)                    # the active size is sticky context state: do not leak it to later calls
        return res, n

    # ------------------------------------------------------------------ visual-odometry path (TwistNode)
    def vo_match(self, desc_q, n_q, desc_r, n_r, ratio: float = 0.7, want_knn: bool = False):
        """gn_vo_match: BFMatcher(L2).knnMatch(k=2) + ratio test.  desc [B,S,128] f32, n [B] i32 (device).
        Returns (idx [B,kmax,2] i64, dist [B,kmax] f32, n_good [B] i32[, nn_idx [B,kmax,2] i32, nn_dist [B,kmax,2] f32])."""
        B = desc_q.shape[0]
        idx = torch.empty((B, self.kmax, 2), dtype=torch.int64, device=self.device)
        dist = torch.empty((B, self.kmax), dtype=torch.float32, device=self.device)
        n_good = torch.empty((B,), dtype=torch.int32, device=self.device)
        nn_idx = torch.empty((B, self.kmax, 2), dtype=torch.int32, device=self.device) if want_knn else None
        nn_dist = torch.empty((B, self.kmax, 2), dtype=torch.float32, device=self.device) if want_knn else None
        rc = self.lib.gn_vo_match(self.ctx, B, _ptr(desc_q), _ptr(n_q), desc_q.shape[1], _ptr(desc_r), _ptr(n_r), desc_r.shape[1],
                                  float(ratio), _ptr(idx), _ptr(dist), _ptr(n_good), _ptr(nn_idx), _ptr(nn_dist), self._stream())
        _lib.check(self.ctx, rc, "gn_vo_match")
        return (idx, dist, n_good, nn_idx, nn_dist) if want_knn else (idx, dist, n_good)

    def vo_estimate(self, inputs: dict, K: np.ndarray, ratio: float = 0.7, min_matches: int = 30, out: Optional[dict] = None):
        """gn_vo_estimate on staged inputs: TwistNode._pose lines 227-289 for the whole batch."""
        B = inputs["desc_q"].shape[0]
        if out is None:
            out = self.alloc_outputs(B)
        K9 = np.ascontiguousarray(np.asarray(K, np.float64).reshape(9))
        rc = self.lib.gn_vo_estimate(self.ctx, B, inputs["kpt_format"],
                                     _ptr(inputs["desc_q"]), _ptr(inputs["kpt_q"]), _ptr(inputs["n_q"]), inputs["desc_q"].shape[1],
                                     _ptr(inputs["desc_r"]), _ptr(inputs["kpt_r"]), _ptr(inputs["n_r"]), inputs["desc_r"].shape[1],
                                     K9.ctypes.data_as(_lib.c_f64p), float(ratio), min_matches,
                                     _ptr(out["R"]), _ptr(out["t"]), _ptr(out["n_match"]), _ptr(out["n_inliers"]), _ptr(out["ok"]),
                                     self._stream())
        _lib.check(self.ctx, rc, "gn_vo_estimate")
        return out

    def vo_estimate_images(self, frames_q, frames_r, K: np.ndarray, sift=None, ratio: float = 0.7, min_matches: int = 30, out: Optional[dict] = None):
        """TwistNode._pose from pixels for B frame pairs (twist_node.py:227-289), everything in HBM: one batched SIFT pass over
        the B query frames and the B reference (previous) frames, then brute-force 2-NN + ratio test + planar PnP
        (`gn_vo_estimate`).  Returns (outputs, n_keypoints [2B] int32 on the host: query frames first)."""
        from .sift import SIFT
        if sift is None:
            if getattr(self, "_sift", None) is None:
                self._sift = SIFT(engine=self, max_keypoints=self.kmax)
            sift = self._sift
        to_dev = lambda a: a if isinstance(a, torch.Tensor) else torch.as_tensor(np.ascontiguousarray(a, np.uint8), device=self.device)  # noqa: E731
        q, r = to_dev(frames_q), to_dev(frames_r)
        assert q.shape == r.shape and q.dim() == 3, "expected two (B, H, W) uint8 stacks of one size"
        B = int(q.shape[0])
        kpt, _, _, desc, n = sift.detect_and_compute_batch_device(torch.cat([q, r], 0))
        nd = torch.as_tensor(n, device=self.device)
        inputs = dict(desc_q=desc[:B], kpt_q=kpt[:B], n_q=nd[:B], desc_r=desc[B:], kpt_r=kpt[B:], n_r=nd[B:], kpt_format=_lib.GN_KPT_XYSA)
        return self.vo_estimate(inputs, K, ratio, min_matches, out=out), n

    def alloc_outputs(self, B: int) -> dict:
        d = self.device
        return dict(R=torch.empty((B, 3, 3), dtype=torch.float64, device=d), t=torch.empty((B, 3, 1), dtype=torch.float64, device=d),
                    n_match=torch.empty((B,), dtype=torch.int32, device=d), n_inliers=torch.empty((B,), dtype=torch.int32, device=d),
                    ok=torch.empty((B,), dtype=torch.uint8, device=d))

    # ------------------------------------------------------------------ test hooks
    def debug_read(self, name: str, count: int, dtype=np.float32) -> np.ndarray:
        buf = np.empty(count, dtype=dtype)
        n = self.lib.gn_debug_read(self.ctx, name.encode(), buf.ctypes.data_as(C.c_void_p), buf.nbytes, self._stream())
        _lib.check(self.ctx, int(n), f"gn_debug_read({name})")
        return buf[: int(n)]

    def set_stage_timing(self, enable: bool) -> None:
        _lib.check(self.ctx, self.lib.gn_set_stage_timing(self.ctx, int(enable)), "gn_set_stage_timing")

    def stage_ms(self) -> Dict[str, float]:
        buf = (C.c_float * 16)()
        n = self.lib.gn_get_stage_ms(self.ctx, buf, 16)
        _lib.check(self.ctx, n, "gn_get_stage_ms")
        return {name: float(buf[i]) for i, name in enumerate(_lib.STAGE_NAMES[:n])}

    def set_kernel_timing(self, max_launches: int) -> None:
        _lib.check(self.ctx, self.lib.gn_set_kernel_timing(self.ctx, max_launches), "gn_set_kernel_timing")

    def kernel_stats(self, kernel_class: int = 0) -> Dict[str, float]:
        """HIP-event totals of the recorded launches; kernel_class 0 = GEMMs, 1 = attention."""
        buf = (C.c_double * 3)()
        _lib.check(self.ctx, self.lib.gn_get_kernel_stats(self.ctx, kernel_class, buf), "gn_get_kernel_stats")
        by = C.c_double(0.0)
        _lib.check(self.ctx, self.lib.gn_get_kernel_bytes(self.ctx, kernel_class, C.byref(by)), "gn_get_kernel_bytes")
        return {"launches": buf[0], "ms": buf[1], "flops": buf[2], "bytes": by.value}

    def kernel_table(self):
        """[{name, launches, ms, flops, bytes}] of the launches recorded since set_kernel_timing (HIP events on the launch stream)."""
        import json
        buf = C.create_string_buffer(1 << 16)
        n = self.lib.gn_get_kernel_table(self.ctx, buf, len(buf))
        _lib.check(self.ctx, n, "gn_get_kernel_table")
        return json.loads(buf.value.decode())

    def debug_gemm(self, A: torch.Tensor, W: torch.Tensor, bias: Optional[torch.Tensor]) -> torch.Tensor:
        M, K = A.shape
        N = W.shape[0]
        Y = torch.empty((M, N), dtype=torch.float32, device=self.device)
        rc = self.lib.gn_debug_gemm(self.ctx, M, N, K, _ptr(A), _ptr(W), _ptr(bias), _ptr(Y), self._stream())
        _lib.check(self.ctx, rc, "gn_debug_gemm")
        return Y

    def debug_attention(self, q, k, v, nkv, cross: bool, qscale: float) -> torch.Tensor:
        """q,k,v [BS, npad, 256] f32 (4 heads x 64 concatenated); nkv [BS] i32."""
        BS, npad, _ = q.shape
        out = torch.empty_like(q)
        rc = self.lib.gn_debug_attention(self.ctx, BS, npad, int(cross), qscale, _ptr(q), 256, _ptr(k), 256, _ptr(v), 256,
                                         _ptr(nkv), _ptr(out), 256, self._stream())
        _lib.check(self.ctx, rc, "gn_debug_attention")
        return out
