import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "faster-orefsdet_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    def _load(name):
        return dict(np.load(os.path.join(GOLDEN, name + ".npz")))

    return _load


def rel_err(a, b):
    """max |a-b| / max(|b|): the 'fp32 feature maps within 1e-4 rel' measure used throughout."""
    import numpy as np

    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def chan_err(a, b, axis=1, floor=1e-2):
    """Per-channel companion of rel_err (under the max-norm measure a channel -- or a pyramid level -- whose
    magnitude is 1 % of the tensor's maximum could be 1 % wrong and pass "1e-4 rel").  For every channel c along `axis`:
    rms(a_c - b_c) / max(rms(b_c), floor * rms(b)); the maximum over channels is returned.  The floor (1 % of the whole tensor's rms)
    only keeps channels that are dead after a ReLU from dividing by zero."""
    import numpy as np

    a = np.moveaxis(np.asarray(a, dtype=np.float64), axis, 0)
    b = np.moveaxis(np.asarray(b, dtype=np.float64), axis, 0)
    a, b = a.reshape(a.shape[0], -1), b.reshape(b.shape[0], -1)
    num = np.sqrt(((a - b) ** 2).mean(1))
    den = np.maximum(np.sqrt((b ** 2).mean(1)), floor * np.sqrt((b ** 2).mean()) + 1e-30)
    v = float((num / den).max())
    log = os.environ.get("ORE_CHAN_LOG")
    if log:                                                   # measurement aid: what the per-channel figure actually is, per call site
        import inspect
        fr = inspect.stack()[1]
        with open(log, "a") as f:
            f.write("%s:%d %s chan_err %.3e rel_err %.3e\n" % (os.path.basename(fr.filename), fr.lineno, fr.function, v, rel_err(a, b)))
    return v
