import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, 'd3human-code_amd')
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

EMUL_SO = os.path.join(ROOT, 'tests', 'emul', 'libd3h_emul.so')
GOLD = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
    config.addinivalue_line('markers', 'slow: minutes of host time on the GPU box (the oracle chain at a BASELINE size)')
    if os.environ.get('D3H_TEST_POISON') == '1':
        # debugging aid: torch.empty() returns NaN-filled (float) / max-int memory, so a kernel that reads an output buffer it was supposed
        # to write completely, or a wrapper that forgets a zero fill, shows up as NaN / a wild index instead of passing by luck
        import torch
        torch.use_deterministic_algorithms(True, warn_only=True)
        torch.utils.deterministic.fill_uninitialized_memory = True


def _emul_stale():
    if not os.path.exists(EMUL_SO):
        return True
    t = os.path.getmtime(EMUL_SO)
    import glob
    srcs = glob.glob(os.path.join(PKG, 'csrc', '*.hip')) + glob.glob(os.path.join(PKG, 'csrc', '*.h')) + \
        [os.path.join(ROOT, 'tests', 'emul', 'hip_emul.h')]
    return any(os.path.getmtime(s) > t for s in srcs)


@pytest.fixture(scope='session')
def emul_lib():
    """Host emulation of the kernel sources (tests/emul/hip_emul.h) -- kernel-logic debugging only."""
    if _emul_stale():
        subprocess.check_call([os.path.join(ROOT, 'tests', 'emul', 'build_emul.sh')], stdout=subprocess.DEVNULL,
                              stderr=subprocess.DEVNULL)
    return EMUL_SO


@pytest.fixture()
def emul(emul_lib):
    from d3h import _lib as L
    L._use_emulator_for_tests(emul_lib)
    yield 'cpu'
    L._lib = None
    L._emulated = False


@pytest.fixture()
def gpu():
    import torch
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    from d3h import _lib as L
    L._lib = None
    L._emulated = False
    return 'cuda'


def golden(name):
    import numpy as np
    return np.load(os.path.join(GOLD, name))
