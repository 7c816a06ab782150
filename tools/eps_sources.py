"""Where the fast mode's eps comes from (developer measurement): gn_calibrate_certify on 16 x 1024 with the attention input projections on
two (shipped) and three partial products (knob 27), on the three-product block tail.  Prints measured max |P_mode - P_f32| per weight family."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gisnav_amd.engine import PoseEngine  # noqa: E402
from gisnav_amd.synthetic import make_pair  # noqa: E402
from gisnav_amd.weights import default_init_state_dict, synthetic_state_dict  # noqa: E402

MID = dict(ffn_out_std=1.2e-3, final_scale=12.0, matchability_bias=2.0, matchability_std=0.05)
FAMS = {"margin_built": (synthetic_state_dict(0), 0.5), "mid_margin": (synthetic_state_dict(0, **MID), 0.01), "default_init": (default_init_state_dict(0), 0.0)}
cal_pairs = [make_pair(4460 + i, n_q=1024, n_r=1000) for i in range(16)]
for name, (sd, th) in FAMS.items():
    for qkv in (2, 3):
        eng = PoseEngine(0, max_batch=16, max_kpts=1024, precision="f16x2_f16_attn", state_dict=sd, filter_threshold=th)
        eng.lib.gn_debug_set_variant(eng.ctx, 27, qkv)
        cal = eng.calibrate_certify(eng.stage_inputs(cal_pairs), safety=1.0)
        print(f"{name:14s} qkv products {qkv}: max |dP| = {cal['measured']:.3e}", flush=True)
        del eng
