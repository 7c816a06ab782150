"""Golden fixtures for the LightGlue(features="superpoint") matcher, generated with THIRD-PARTY code that is importable in the build
container: transformers' LightGlueForKeypointMatching (a SuperPoint-LightGlue port) run on CPU with the repo's seeded synthetic
weights mapped into it (its separate cross-attention q / k projections both take the shared ``to_qk``).  Unlike the SIFT variant
(kornia / cv2 absent) this fixture does NOT come from the repo's own oracle.

    python tests/golden/make_superpoint_golden.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from gisnav_amd.synthetic import make_pair_256  # noqa: E402
from gisnav_amd.weights import synthetic_state_dict  # noqa: E402
from test_oracle_pins import _hf_layer  # noqa: E402


def hf_model(sd, filter_threshold):
    from transformers.models.lightglue.configuration_lightglue import LightGlueConfig
    from transformers.models.lightglue.modeling_lightglue import LightGlueForKeypointMatching
    tsd = {k: torch.from_numpy(v) for k, v in sd.items()}
    cfg = LightGlueConfig(descriptor_dim=256, num_hidden_layers=9, num_attention_heads=4, depth_confidence=-1.0, width_confidence=-1.0,
                          filter_threshold=filter_threshold)
    cfg._attn_implementation = "eager"
    hf = LightGlueForKeypointMatching(cfg).eval()
    hf.positional_encoder.projector.weight.data = tsd["posenc.Wr.weight"]
    for i in range(9):
        hf.transformer_layers[i].load_state_dict(_hf_layer(sd, i).state_dict())
        hf.match_assignment_layers[i].final_projection.weight.data = tsd[f"log_assignment.{i}.final_proj.weight"]
        hf.match_assignment_layers[i].final_projection.bias.data = tsd[f"log_assignment.{i}.final_proj.bias"]
        hf.match_assignment_layers[i].matchability.weight.data = tsd[f"log_assignment.{i}.matchability.weight"]
        hf.match_assignment_layers[i].matchability.bias.data = tsd[f"log_assignment.{i}.matchability.bias"]
    return hf


def main():
    import transformers
    torch.set_num_threads(1)
    sd = synthetic_state_dict(0, feature="superpoint")
    hf = hf_model(sd, 0.1)
    for name, (seed, n, h, w) in {"lightglue_sp_seed0_n200_640x480": (31, 200, 480, 640), "lightglue_sp_seed0_n384_1920x1080": (32, 384, 1080, 1920)}.items():
        p = make_pair_256(seed, n_q=n, n_r=n, h=h, w=w)
        kq, kr = torch.from_numpy(p.kp_q), torch.from_numpy(p.kp_r)
        dq, dr = torch.from_numpy(p.desc_q), torch.from_numpy(p.desc_r)
        with torch.inference_mode():
            out = hf._match_image_pair(torch.stack([kq, kr])[None], torch.stack([dq, dr])[None], h, w, mask=torch.ones(1, 2, n, dtype=torch.int),
                                       output_hidden_states=True)
        m0, s0 = out[0].reshape(-1, n)[0], out[1].reshape(-1, n)[0]
        valid = m0 > -1
        idx = torch.stack([torch.where(valid)[0], m0[valid].long()], -1).numpy()
        x_final = out[3][-3].numpy()          # descriptors after the last layer, (2, n, 256)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), kp_q=p.kp_q, desc_q=p.desc_q, kp_r=p.kp_r, desc_r=p.desc_r, h=h, w=w,
                            idx=idx, scores=s0[valid].numpy(), x_final=x_final, source=f"transformers {transformers.__version__} LightGlueForKeypointMatching, CPU fp32")
        print(name, "matches", len(idx))


if __name__ == "__main__":
    main()
