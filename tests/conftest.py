"""pytest configuration: `gpu` marker, repo root on sys.path, shared fixtures."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault('NAFP_TEST_HOOKS', '1')      # the library's test hook (option 6, tests/test_gpu_backward.py) is refused in any other process
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
if os.path.join(ROOT, 'tests') not in sys.path:
    sys.path.insert(0, os.path.join(ROOT, 'tests'))


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def pytest_collection_modifyitems(config, items):
    """GPU tests fail loudly on a box without a GPU only if explicitly selected
    with -m gpu; otherwise they are skipped."""
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason='no GPU in this container')
    for item in items:
        if 'gpu' in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope='session')
def nafp():
    import neural_audio_fp_amd
    return neural_audio_fp_amd


@pytest.fixture(scope='session')
def cfg():
    import yaml
    with open(os.path.join(ROOT, 'config', 'default.yaml')) as f:
        return yaml.safe_load(f)


@pytest.fixture(params=['f32', 'x6'])
def arith(request, monkeypatch):
    """The arithmetic of the inference forward: 'f32' = the fp32 MFMA path (the default and the headline), 'x6' = the exact 3-way
    bf16 split with six products (NAFP_OPT_BF16X3 = 2: float32-equivalent products on the bf16 matrix pipe).  Selected through the
    environment (NAFP_BF16X3, read by FingerPrinter.__init__) so that every model the test -- or a child process it starts --
    creates runs on it.  Every forward test that compares with the oracle takes this fixture: SAME tolerances for both
    (the promotion gate of DESIGN.md section 9)."""
    if request.param == 'x6':
        monkeypatch.setenv('NAFP_BF16X3', '2')
    else:
        monkeypatch.delenv('NAFP_BF16X3', raising=False)
    return request.param


@pytest.fixture(scope='session')
def golden():
    """Vectors written by tests/gen_golden.py from the float64 oracle (not from the
    reference: it cannot run here; see oracle/__init__.py)."""
    import numpy as np
    return dict(np.load(os.path.join(ROOT, 'tests', 'golden', 'hotpath_v1.npz')))


# ---- observed maxima next to their tolerances: printed in the terminal summary even under -q, so the
# driver's GPUTEST tail carries them, and kept in gpurun_out/observed_tolerances.json ----
_OBSERVED = {}


@pytest.fixture
def observe(request):
    """observe(what, value, tol): records max(value) per (test, what) and asserts value < tol."""
    def rec(what, value, tol):
        key = f'{request.node.name}::{what}'
        value = float(value)
        cur = _OBSERVED.get(key)
        _OBSERVED[key] = (max(value, cur[0]) if cur else value, float(tol))
        assert value < tol, (what, value, tol)
    return rec


def pytest_terminal_summary(terminalreporter):
    if not _OBSERVED:
        return
    terminalreporter.write_line('observed error maxima (value / tolerance):')
    for k in sorted(_OBSERVED):
        v, t = _OBSERVED[k]
        terminalreporter.write_line(f'  {k}: {v:.3e} / {t:.1e}' + (f'  ({t / v:.0f}x headroom)' if v > 1e-30 else ''))
    try:
        import json
        os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
        with open(os.path.join(ROOT, 'gpurun_out', 'observed_tolerances.json'), 'w') as f:
            json.dump({k: {'observed': v, 'tolerance': t} for k, (v, t) in _OBSERVED.items()}, f, indent=1, sort_keys=True)
    except OSError:
        pass
