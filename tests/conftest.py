import glob
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def golden(name):
    """Load one fixture as a dict of numpy arrays."""
    with np.load(os.path.join(GOLDEN_DIR, name + ".npz")) as z:
        return {k: z[k] for k in z.files}


def render_fixture_names():
    """All fixtures that record one HelioField.render call."""
    out = []
    for p in sorted(glob.glob(os.path.join(GOLDEN_DIR, "g*.npz"))):
        n = os.path.basename(p)[:-4]
        if n.startswith(("g6_", "g9_")):
            continue
        out.append(n)
    return out


@pytest.fixture(scope="session")
def golden_loader():
    return golden


# HelioEnv fixtures recorded from the reference (tests/golden/make_golden.py, make_env_golden):
# tag → (file stem, use_error_mask, exponential_risk, single_sun, azimuth, elevation)
ENV_FIXTURES = {
    "train": ("g6_env_train_n50_b25_r64", False, False, False, 45.0, 45.0),
    "readme": ("g6_env_readme_n50_b25_r64", False, False, False, 45.0, 45.0),
    "mask": ("g6_env_mask_n50_b25_r64", True, False, False, 45.0, 45.0),
    "exprisk": ("g6_env_exprisk_n20_b12_r48", False, True, False, 45.0, 45.0),
    "single": ("g6_env_single_n20_b12_r48", True, False, True, 30.0, 50.0),
}
