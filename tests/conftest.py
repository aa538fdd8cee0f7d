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


# ---------------------------------------------------------------------------------------------------------
# Gradient bars.  Every gradient comparison of the GPU tests goes through check_grad(): the deviation
# max|got - ref| / max|ref| is held to a bar that is at most 2x the worst deviation MEASURED for that test on
# an MI355X (tests/grad_floor.py, profiles/r04_a_grad_floor.txt and profiles/r04_b_grad_devs.txt — the latter is
# what HELIO_RECORD_DEVS=<file> makes this helper write: one line per call with the measured value and its bar).
# The measured floor is 1e-7 … 1e-6 (a few fp32 roundings of the largest entry); north_star's bar is 1e-5.
# ---------------------------------------------------------------------------------------------------------
def grad_deviation(got, ref):
    got, ref = np.asarray(got, dtype=np.float64), np.asarray(ref, dtype=np.float64)
    return float(np.abs(got.reshape(ref.shape) - ref).max() / max(np.abs(ref).max(), 1e-300))


def check_grad(got, ref, bar, tag=""):
    """Assert max|got - ref| <= bar * max|ref|; tensors or arrays.  Returns the measured deviation."""
    to_np = lambda t: t.detach().cpu().numpy() if hasattr(t, "detach") else np.asarray(t)  # noqa: E731
    dev = grad_deviation(to_np(got), to_np(ref))
    out = os.environ.get("HELIO_RECORD_DEVS")
    if out:
        test = os.environ.get("PYTEST_CURRENT_TEST", "?").split(" ")[0]
        with open(out, "a") as fh:
            fh.write(f"{dev:.3e} bar {bar:.1e}  {test} {tag}\n")
    assert dev <= bar, f"{tag}: gradient deviates by {dev:.3e} of max|ref| (bar {bar:.1e})"
    return dev
