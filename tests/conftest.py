import importlib
import os
import sys

# Launch shapes: the measured autotuner picks whatever is fastest on the day, i.e. a timing-dependent summation
# order.  The suite runs on the deterministic heuristic shapes instead (every tile shape is forced explicitly by the
# launch-configuration tests, and the tuner itself by test_autotuner_pins_a_valid_configuration).
os.environ.setdefault('GCA_AUTOTUNE', '0')

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def pytest_collection_modifyitems(config, items):
    # gpu-marked tests are skipped (not failed) when no device is present and -m gpu was not forced
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason='no GPU in this container')
    for it in items:
        if 'gpu' in it.keywords:
            it.add_marker(skip)


class Golden:
    """Lazy view of one tests/golden/*.npz with ':'-prefixed groups."""

    def __init__(self, name):
        self.z = np.load(os.path.join(GOLDEN, name + '.npz'), allow_pickle=False)

    def t(self, key):
        a = self.z[key]
        return torch.from_numpy(np.array(a))

    def group(self, prefix):
        return {k[len(prefix):]: torch.from_numpy(np.array(self.z[k])) for k in self.z.files if k.startswith(prefix)}

    def keys(self):
        return list(self.z.files)

    def x(self, key):
        """Regenerate a seeded input from its stored (seed, *shape) spec (see make_golden.seeded_randn)."""
        sp = [int(v) for v in self.z[key]]
        return torch.randn(*sp[1:], generator=torch.Generator().manual_seed(sp[0]))


@pytest.fixture(scope='session')
def golden():
    cache = {}

    def get(name):
        if name not in cache:
            cache[name] = Golden(name)
        return cache[name]
    return get


@pytest.fixture(scope='session')
def pkg():
    """The product package (directory name has a hyphen, so import it by string)."""
    return importlib.import_module('video-graph-ssl_amd')


def rel_err(a, b):
    """max |a-b| / max(|b|) -- the 'relative fp32' measure used for the 1e-3 bar."""
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


@pytest.fixture(autouse=True)
def _release_gpu_objects():
    """Drop hipGraphs / trainers of a test before the next one starts (and long before interpreter shutdown: a
    captured graph destroyed after the HIP runtime has been torn down aborts the process at exit)."""
    yield
    import gc
    gc.collect()
    if torch.cuda.is_available():
        torch.cuda.synchronize()
