"""Host-side mirror of `kornia.feature.LoFTR` -- the detector-free matcher BASELINE.json's north_star / configs[1] name ("LoFTR matcher on
1 MI355X, HIP conv + attention kernels, fp32").  The reference tree no longer contains it (only the word: docs/vitepress/docs/glossary.md:186);
older GISNav releases called `LoFTR(pretrained="outdoor")({"image0": ..., "image1": ...})` and read `keypoints0`, `keypoints1`, `confidence`
from the result -- the call signature and output dictionary mirrored here.  Marshalling only: backbone, linear-attention transformer,
dual-softmax coarse matching and the fine level run in libgisnav_amd.so (`gn_loftr_*`, csrc/gn_loftr.hip)."""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional

import numpy as np
import torch

from . import _lib


class LoFTR:
    """`LoFTR(state_dict=...).to("cuda:0").eval()`; `out = m({"image0": img0, "image1": img1})` with (1, 1, H, W) or (H, W) float images in
    [0, 1] of equal size (H, W multiples of 8) -> {"keypoints0" (M, 2), "keypoints1" (M, 2), "confidence" (M,), "batch_indexes" (M,)} on the
    input device, matches in ascending coarse cell of image0.  `fine=False` stops after the coarse level (keypoints on the 1/8 grid)."""

    def __init__(self, pretrained: Optional[str] = None, *, state_dict: Optional[Dict] = None, max_matches: Optional[int] = None, fine: bool = True, graph: bool = True,
                 arithmetic: str = "exact_f32"):
        if state_dict is None:
            state_dict = self._find_pretrained(pretrained or "outdoor")
        # max_matches None = every mutual match (at most one per coarse cell of image0), as kornia returns them; a number caps the list (first in raster order)
        self._sd, self._max, self._fine, self._graph = state_dict, (None if max_matches is None else int(max_matches)), bool(fine), bool(graph)
        self._arith = {"exact_f32": 0, "split_fp16": 1}[arithmetic]   # split_fp16: f32-accurate 2-term fp16 operands (gn_loftr_set_arithmetic)
        self._ctx, self._shape, self._device = None, None, None
        self.lib = None

    @staticmethod
    def _find_pretrained(name: str):
        """kornia downloads `loftr_{name}.ckpt` through torch.hub; this mirror looks in the same cache directory (and in
        $GISNAV_AMD_LOFTR_WEIGHTS) but never downloads."""
        import os
        cands = [os.environ.get("GISNAV_AMD_LOFTR_WEIGHTS")]
        try:
            cands.append(os.path.join(torch.hub.get_dir(), "checkpoints", f"loftr_{name}.ckpt"))
        except Exception:  # noqa: BLE001
            pass
        for c in cands:
            if c and os.path.exists(c):
                sd = torch.load(c, map_location="cpu", weights_only=True)
                return sd.get("state_dict", sd)
        return None

    def to(self, device):
        device = torch.device(device)
        if device.type != "cuda":
            raise _lib.GnError("gisnav_amd.LoFTR runs on an MI355X only (no CPU path)")
        if self._sd is None:
            raise _lib.GnError("no LoFTR weights: kornia would download loftr_outdoor.ckpt here; offline, put the checkpoint into torch.hub's "
                               "checkpoints directory, set GISNAV_AMD_LOFTR_WEIGHTS, or pass state_dict=")
        self._device = device
        self.lib = _lib.load()
        return self

    def eval(self):
        return self

    def __del__(self):
        ctx, self._ctx = getattr(self, "_ctx", None), None
        if ctx and self.lib is not None:
            self.lib.gn_loftr_destroy(ctx)

    def _check(self, rc: int, what: str) -> None:
        if rc < 0:
            msg = self.lib.gn_loftr_last_error(self._ctx).decode() if self._ctx else self.lib.gn_loftr_last_error(None).decode()
            raise _lib.GnError(f"{what} failed ({rc}): {msg}")

    def _cap(self, H: int, W: int) -> int:
        L = (H // 8) * (W // 8)
        return min(L if self._max is None else min(self._max, L), 131072)     # (gn_loftr_create's own bound)

    def _ensure(self, H: int, W: int) -> None:
        if self._ctx is not None and self._shape == (H, W):
            return
        if self._ctx is not None:
            self.lib.gn_loftr_destroy(self._ctx)
            self._ctx = None
        ctx = C.c_void_p()
        rc = self.lib.gn_loftr_create(self._device.index or 0, H, W, self._cap(H, W), int(self._fine), C.byref(ctx))
        if rc < 0:
            raise _lib.GnError(f"gn_loftr_create failed ({rc}): {self.lib.gn_loftr_last_error(None).decode()}")
        self._ctx, self._shape = ctx, (H, W)
        self.lib.gn_loftr_set_graph(ctx, int(self._graph))
        self.lib.gn_loftr_set_arithmetic(ctx, self._arith)
        for name, arr in self._sd.items():
            if hasattr(arr, "detach"):
                arr = arr.detach().cpu().numpy()
            if name.startswith("matcher."):            # kornia's checkpoint nests the model under `matcher.`
                name = name[len("matcher."):]
            if name.endswith("num_batches_tracked") or name == "pos_encoding.pe" or (not self._fine and (name.startswith("loftr_fine") or name.startswith("fine_preprocess"))):
                continue
            arr = np.ascontiguousarray(arr, dtype=np.float32)
            shape = (C.c_int64 * max(arr.ndim, 1))(*(arr.shape if arr.ndim else (1,)))
            self._check(self.lib.gn_loftr_load_tensor(ctx, name.encode(), arr.ctypes.data_as(C.c_void_p), shape, max(arr.ndim, 1)), f"gn_loftr_load_tensor({name})")
        missing = self.lib.gn_loftr_missing_tensors(ctx)
        if missing:
            raise _lib.GnError(f"{missing} required LoFTR tensors missing from the state dict")

    @torch.inference_mode()
    def __call__(self, data: Dict[str, torch.Tensor], with_ids: bool = False) -> Dict[str, torch.Tensor]:
        if self._device is None:
            raise _lib.GnError("call .to(device) first")
        i0, i1 = data["image0"], data["image1"]
        if i0.shape != i1.shape:
            raise _lib.GnError("image0 and image1 must have one size (the published model pads / masks otherwise; not built)")
        in_dev = i0.device
        f = lambda t: t.to(device=self._device, dtype=torch.float32).reshape(t.shape[-2], t.shape[-1]).contiguous()  # noqa: E731
        a, b = f(i0), f(i1)
        H, W = int(a.shape[0]), int(a.shape[1])
        self._ensure(H, W)
        M = self._cap(H, W)
        k0 = torch.empty((M, 2), dtype=torch.float32, device=self._device); k1 = torch.empty_like(k0)
        conf = torch.empty((M,), dtype=torch.float32, device=self._device)
        ij = torch.empty((M, 2), dtype=torch.int32, device=self._device)
        n = C.c_int32(0)
        stream = C.c_void_p(torch.cuda.current_stream(self._device).cuda_stream)
        p = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
        self._check(self.lib.gn_loftr_match(self._ctx, p(a), p(b), p(k0), p(k1), p(conf), p(ij), C.byref(n), stream), "gn_loftr_match")
        m = int(n.value)
        out = {"keypoints0": k0[:m].to(in_dev), "keypoints1": k1[:m].to(in_dev), "confidence": conf[:m].to(in_dev),
               "batch_indexes": torch.zeros(m, dtype=torch.int64, device=in_dev)}
        if with_ids:
            out["i_ids"], out["j_ids"] = ij[:m, 0].to(in_dev).long(), ij[:m, 1].to(in_dev).long()
        return out

    forward = __call__

    def debug_read(self, name: str, count: int) -> np.ndarray:
        buf = np.empty(count, dtype=np.float32)
        n = self.lib.gn_loftr_debug_read(self._ctx, name.encode(), buf.ctypes.data_as(C.c_void_p), buf.nbytes, C.c_void_p(torch.cuda.current_stream(self._device).cuda_stream))
        if n < 0:
            raise _lib.GnError(f"gn_loftr_debug_read({name}) failed ({n})")
        return buf[: int(n)]


def loftr_pose(matcher: "LoFTR", engine, frame01: torch.Tensor, tile01: torch.Tensor, dem, K, min_matches: int = 15, conf_threshold: float = 0.0):
    """Camera frame <-> map tile pose with the detector-free matcher in front of the SAME solver as the SIFT / LightGlue path: LoFTR matches
    (`keypoints0` in the frame, `keypoints1` in the tile) -> DEM lift of the tile points (`_shared.py:95-102`) -> solvePnPRansac + Rodrigues
    (`gn_gather_points`, `gn_pnp_ransac`; seam B2).  Everything stays on the device.  frame01 / tile01: (H, W) float in [0, 1];
    dem: (H, W) uint8 or None.  Returns (R (3,3), t (3,1), n_matches) or None below `min_matches` / when RANSAC finds no model."""
    out = matcher({"image0": frame01, "image1": tile01})
    keep = out["confidence"] > conf_threshold
    k0, k1 = out["keypoints0"][keep], out["keypoints1"][keep]
    n = int(k0.shape[0])
    if n < min_matches or n > engine.kmax:
        if n > engine.kmax:
            engine.grow(((n + 1023) // 1024) * 1024)
        if n < min_matches:
            return None
    dev = engine.device
    pad = lambda k: torch.cat([k.to(dev), torch.zeros((n, 2), dtype=torch.float32, device=dev)], 1).reshape(1, n, 4).contiguous()  # noqa: E731  GN_KPT_XYSA rows
    idx = torch.arange(n, dtype=torch.int64, device=dev).repeat_interleave(2).reshape(1, n, 2)
    idx_full = torch.zeros((1, engine.kmax, 2), dtype=torch.int64, device=dev)
    idx_full[0, :n] = idx[0]
    nm = torch.tensor([n], dtype=torch.int32, device=dev)
    d = None if dem is None else torch.as_tensor(np.ascontiguousarray(dem, np.uint8), device=dev)[None]
    mkp, obj = engine.gather_points(pad(k0), pad(k1), idx_full, nm, d, _lib.GN_KPT_XYSA)
    R, t, n_inl, ok = engine.pnp_ransac(obj, mkp, nm, np.asarray(K, np.float64).reshape(3, 3), min_pts=min_matches)
    if not bool(ok.cpu()[0]):
        return None
    return R[0].cpu().numpy(), t[0].cpu().numpy(), n
