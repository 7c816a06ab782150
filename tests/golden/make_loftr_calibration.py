"""Produces gisnav_amd/data/loftr_synth_calib_seed0.npz: the three mean backbone-feature vectors the synthetic LoFTR weights are calibrated
with (oracle/loftr.py::synthetic_state_dict does the same computation in place).  Run once; the product-side generator
(gisnav_amd/loftr_synthetic.py) reads the vectors as data and tests assert that both generators give identical tensors.

    python tests/golden/make_loftr_calibration.py
"""
import inspect
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import loftr as lf  # noqa: E402


def main():
    code = inspect.getsource(lf).replace("    with torch.inference_mode():\n        img, _ = synthetic_pair(12345, 96, 128)",
                                         "    if False:\n        img, _ = synthetic_pair(12345, 96, 128)")
    ns = {}
    exec(compile(code, "loftr_uncalibrated", "exec"), ns)      # the generator without its calibration step
    sd0 = ns["synthetic_state_dict"](0)
    with torch.inference_mode():
        img, _ = lf.synthetic_pair(12345, 96, 128)
        taps = {}
        lf.backbone(sd0, img[None, None], taps)
        means = {k: taps[t].mean((0, 2, 3)).numpy() for k, t in (("layer3_outconv", "x3"), ("layer2_outconv", "x2"), ("layer1_outconv", "x1"))}
    np.savez(os.path.join(ROOT, "gisnav_amd", "data", "loftr_synth_calib_seed0.npz"), **means)


if __name__ == "__main__":
    main()
