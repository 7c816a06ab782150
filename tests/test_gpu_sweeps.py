"""The sweeps of tests/sweeps/ as part of the GPU suite (-m gpu): inputs OUTSIDE the parametrised cases of the other modules -- random ragged
batches, image sizes from 32 x 32 to 360 x 488 (LoFTR / SuperPoint), odd and extreme aspect ratios (SIFT), random tiles / crops / angles (stereo),
few-point / outlier-heavy / degenerate PnP scenes -- each against its oracle, each script its own verdict (exit code).  Run one by hand with more
trials or larger sizes: `python tests/sweeps/fuzz_loftr_sizes.py 960x1280 1080x1920`."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("script,args", [("fuzz_ragged.py", ["8", "256"]), ("fuzz_loftr_sizes.py", []), ("fuzz_sift.py", []), ("fuzz_superpoint.py", []),
                                         ("fuzz_pnp.py", []), ("fuzz_stereo.py", ["40"])])
def test_sweep_against_the_oracle(script, args):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "sweeps", script), *args], cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-1500:] + "\n" + r.stderr[-800:])
