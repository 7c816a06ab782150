"""Developer tool: time the attention launches of a 32-pair step for several kernel variants / ablations."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gisnav_amd.engine import PoseEngine  # noqa: E402
from gisnav_amd.synthetic import K_MATRIX, make_pair  # noqa: E402
from gisnav_amd.weights import synthetic_state_dict  # noqa: E402


def main():
    B = 32
    eng = PoseEngine(0, max_batch=B, max_kpts=1024, precision="f32x3_bf16_attn", state_dict=synthetic_state_dict(0))
    inp = eng.stage_inputs([make_pair(i, n_q=1024, n_r=1024) for i in range(B)])
    out = eng.alloc_outputs(B)
    for v in [int(a) for a in sys.argv[1:]] or [4, 43, 41]:
        eng.lib.gn_debug_set_variant(eng.ctx, 1, v)
        for _ in range(2):
            eng.estimate(inp, K_MATRIX, out=out)
        eng.set_kernel_timing(93 * 5)
        for _ in range(5):
            eng.estimate(inp, K_MATRIX, out=out)
        torch.cuda.synchronize()
        st = eng.kernel_stats(1)
        eng.set_kernel_timing(0)
        print(f"attn variant {v}: {st['ms'] * 1e3 / st['launches']:.1f} us/launch  {st['flops'] / st['ms'] / 1e9:.0f} TF", flush=True)


if __name__ == "__main__":
    main()
