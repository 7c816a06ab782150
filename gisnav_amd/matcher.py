"""Seam B1: a drop-in for the matcher object PoseNode builds and calls.

    self._matcher = LightGlueMatcher("sift", params={...}).to(device).eval()   pose_node.py:109-121
    dists, idx = self._matcher(desc_q, desc_r, lafs_q, lafs_r)                 pose_node.py:285-287

Same constructor arguments, same call signature, same outputs: `dists (K,1)` float tensor and
`idx (K,2)` int64 tensor on the input device, rows in ascending query index; empty (0,1)/(0,2)
when either side has fewer than 2 descriptors.  All arithmetic runs in libgisnav_amd.so.
"""
from __future__ import annotations

from typing import Dict, Optional

import torch

from . import _lib
from .engine import PoseEngine
from .weights import N_LAYERS

_DEFAULTS = {"n_layers": N_LAYERS, "filter_threshold": 0.1, "depth_confidence": 0.95, "width_confidence": 0.99}


class LightGlueMatcher:
    def __init__(self, feature_name: str = "sift", params: Optional[Dict] = None, *,
                 state_dict=None, max_kpts: int = 4096, precision: str = "f32", certify: bool = True, certify_calibration_calls: int = 8):
        if feature_name not in ("sift", "superpoint"):
            raise NotImplementedError("LightGlue('sift') is what PoseNode uses (pose_node.py:110); 'superpoint' (256-d descriptors, BASELINE configs[4]) "
                                      "is the other variant built here")
        self.feature_name = feature_name
        p = dict(_DEFAULTS)
        p.update(params or {})
        if p["depth_confidence"] > 0 or p["width_confidence"] > 0:
            # the adaptive depth/width branch is dead code in the reference (torch.device == str is
            # always False, pose_node.py:88); only the exhaustive configuration is implemented
            raise NotImplementedError("only depth_confidence = width_confidence = -1 is supported")
        self.params = p
        if state_dict is None:
            state_dict = self._find_pretrained(feature_name)    # kornia loads the pretrained checkpoint in its constructor (pose_node.py:109-121 passes none)
        self._state_dict = state_dict
        self._max_kpts, self._precision = max_kpts, precision
        # fast precision modes: correspondence indices certified against the exact-f32 arithmetic (gn_set_certify: pairs with a decision inside
        # the calibrated error margin are re-run in f32 before the call returns).  eps is calibrated on the first calls' own inputs (each of them
        # is matched a second time in f32: 4 x the largest difference seen), then frozen.
        self._certify = bool(certify) and precision != "f32"
        self._cal_left, self._cal_eps = int(certify_calibration_calls), 0.0
        self._engine: Optional[PoseEngine] = None

    @staticmethod
    def _find_pretrained(feature_name: str):
        """kornia's LightGlue constructor fetches `{feature}_lightglue` (cvg/LightGlue release v0.1_arxiv) through torch.hub into
        <hub_dir>/checkpoints/ and loads it; this mirror looks in the same place (and in $GISNAV_AMD_LIGHTGLUE_WEIGHTS) but never
        downloads.  Returns a state dict or None (then .to() fails loudly unless load_state_dict() was called)."""
        import os
        cands = [os.environ.get("GISNAV_AMD_LIGHTGLUE_WEIGHTS")]
        try:
            hub = os.path.join(torch.hub.get_dir(), "checkpoints")
            cands += [os.path.join(hub, f"{feature_name}_lightglue_v0-1_arxiv-pth"), os.path.join(hub, f"{feature_name}_lightglue_v0-1_arxiv.pth"),
                      os.path.join(hub, f"{feature_name}_lightglue.pth")]
        except Exception:  # noqa: BLE001
            pass
        for c in cands:
            if c and os.path.exists(c):
                return torch.load(c, map_location="cpu", weights_only=True)   # tensors only: never unpickle code from a cache directory
        return None

    # torch.nn.Module-shaped conveniences used by the call site
    def to(self, device):
        device = torch.device(device)
        if device.type != "cuda":
            raise _lib.GnError("gisnav_amd.LightGlueMatcher runs on an MI355X only (no CPU path)")
        if self._state_dict is None:
            raise _lib.GnError(f"no LightGlue('{self.feature_name}') weights: kornia would download {self.feature_name}_lightglue (v0.1_arxiv) here; offline, put the "
                               "checkpoint into torch.hub's checkpoints directory, set GISNAV_AMD_LIGHTGLUE_WEIGHTS, or call load_state_dict() before .to()")
        self._engine = PoseEngine(device.index or 0, max_batch=1, max_kpts=self._max_kpts, precision=self._precision,
                                  state_dict=self._state_dict, n_layers=self.params["n_layers"],
                                  filter_threshold=self.params["filter_threshold"], guard="sync", feature=self.feature_name)
        return self

    def eval(self):
        return self

    def load_state_dict(self, sd):
        self._state_dict = sd
        if self._engine is not None:
            self._engine.load_state_dict(sd)

    @torch.inference_mode()
    def __call__(self, desc1: torch.Tensor, desc2: torch.Tensor, lafs1: torch.Tensor, lafs2: torch.Tensor,
                 hw1=None, hw2=None):
        if self._engine is None:
            raise _lib.GnError("call .to(device) first")
        dev = self._engine.device
        if desc1.shape[0] < 2 or desc2.shape[0] < 2:
            return torch.zeros((0, 1), dtype=desc1.dtype, device=desc1.device), torch.zeros((0, 2), dtype=torch.int64, device=desc1.device)
        f = lambda t: t.to(device=dev, dtype=torch.float32).contiguous()  # noqa: E731
        D = self._engine.desc_dim
        d1, d2 = f(desc1).reshape(1, -1, D), f(desc2).reshape(1, -1, D)
        if max(d1.shape[1], d2.shape[1]) > self._engine.kmax:      # kornia's matcher takes any number of keypoints
            self._engine.grow(((max(d1.shape[1], d2.shape[1]) + 1023) // 1024) * 1024)
        # kornia: image_size = (w, h) from hw when given, else the keypoint extent -- PoseNode passes hw1 = hw2 = None (pose_node.py:285-287).
        # Set on every call: the context normalises THIS call's keypoints by the sizes given with it.
        self._engine.set_image_size(None if hw1 is None else (hw1[1], hw1[0]), None if hw2 is None else (hw2[1], hw2[0]))
        l1, l2 = f(lafs1).reshape(1, -1, 6), f(lafs2).reshape(1, -1, 6)
        n1 = torch.full((1,), d1.shape[1], dtype=torch.int32, device=dev)     # (a fill kernel on the stream, not a blocking pageable upload)
        n2 = torch.full((1,), d2.shape[1], dtype=torch.int32, device=dev)
        fmt = _lib.GN_KPT_LAF | 0x100          # 0x100: descriptors are already normalised by the caller
        if self._certify and self._cal_left > 0:
            try:
                cal = self._engine.calibrate_certify(dict(desc_q=d1, kpt_q=l1, n_q=n1, desc_r=d2, kpt_r=l2, n_r=n2, kpt_format=fmt))
                self._cal_eps = max(self._cal_eps, cal["eps"])
                self._cal_left -= 1
                self._engine.set_certify("rerun", eps=self._cal_eps)
            except _lib.GnError:          # (a sample that cannot calibrate -- it left the fp16 range -- : the next call tries again)
                pass
        idx, score, n_match = self._engine.match(d1, l1, n1, d2, l2, n2, fmt)
        # the D2H sync the reference has at pose_node.py:296-297 -- through a pinned word (pageable reads of a few bytes were measured to stall for
        # ~90 ms every few dozen calls on the MI355X boxes, tools/bench_seams.py)
        if getattr(self, "_n_host", None) is None:
            self._n_host = torch.empty(1, dtype=torch.int32, pin_memory=True)
        self._n_host.copy_(n_match, non_blocking=True)
        torch.cuda.current_stream(dev).synchronize()
        k = int(self._n_host[0])
        return score[0, :k].reshape(-1, 1), idx[0, :k]

    forward = __call__
