import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def state_dict_np():
    from gisnav_amd.weights import synthetic_state_dict
    return synthetic_state_dict(0)


@pytest.fixture(scope="session")
def state_dict_t(state_dict_np):
    import torch
    return {k: torch.from_numpy(v) for k, v in state_dict_np.items()}


def oracle_match(sd_t, p, taps=None, filter_threshold=0.5):
    import torch
    from oracle import lightglue_sift as lg
    tq = torch.from_numpy
    return lg.pose_node_match(sd_t, tq(p.kp_q), tq(p.desc_q), tq(p.size_q), tq(p.angle_q),
                              tq(p.kp_r), tq(p.desc_r), tq(p.size_r), tq(p.angle_r), taps=taps, filter_threshold=filter_threshold)
