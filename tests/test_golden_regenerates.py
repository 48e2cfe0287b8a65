"""tools/gen_golden.py must RUN at HEAD and reproduce the committed fixtures (the three whole-tick generators crashed for
four commits -- oracle/tick.py reached d3h._lib through geometry/perceptual.py -- and nothing noticed).

Dev container only: the generator imports the reference itself (/root/reference; tools/refharness.py) and asserts oracle == reference on
every fixture's inputs while it runs.  Skipped where the reference is absent (the GPU box).  ~75 s."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, 'tests', 'golden')
REF = '/root/reference'


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, 'geometry')), reason='needs the reference checkout (dev container only)')
@pytest.mark.timeout(1200)
def test_all_fixtures_regenerate_from_a_clean_environment(tmp_path):
    env = {k: v for k, v in os.environ.items() if k != 'PYTHONPATH'}          # nothing of the product on the path: the harness strips it anyway
    env['D3H_GOLDEN_OUT'] = str(tmp_path)
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'gen_golden.py')], env=env, capture_output=True, text=True, cwd=str(tmp_path))
    assert r.returncode == 0, r.stdout[-3000:] + '\n' + r.stderr[-3000:]
    committed = sorted(f for f in os.listdir(GOLD) if f.endswith('.npz'))
    made = sorted(f for f in os.listdir(tmp_path) if f.endswith('.npz'))
    assert made == committed and len(made) == 15, (made, committed)
    worst = {}
    for f in committed:
        a, b = np.load(os.path.join(GOLD, f), allow_pickle=False), np.load(os.path.join(tmp_path, f), allow_pickle=False)
        assert sorted(a.files) == sorted(b.files), f
        w = 0.0
        for k in a.files:
            x, y = a[k], b[k]
            assert x.shape == y.shape and x.dtype == y.dtype, (f, k)
            if x.dtype.kind in 'fc':
                assert np.array_equal(np.isfinite(x), np.isfinite(y)), (f, k)
                m = np.isfinite(x)
                if m.any():
                    e = float(np.abs(x[m].astype(np.float64) - y[m].astype(np.float64)).max() / max(1.0, float(np.abs(x[m]).max())))
                    w = max(w, e)
            else:
                assert np.array_equal(x, y), (f, k)               # indices, topology, seeds: bit-exact
        worst[f] = w
    print(worst)
    assert max(worst.values()) <= 1e-7, worst


def test_oracle_package_never_imports_the_product():
    """oracle/ is the checker: importing it (and the trunk module it borrows for the MobileNet-shaped loss) must not pull d3h in"""
    code = ("import sys; sys.path.insert(0, %r); import oracle.tick, oracle.parity, oracle.render, oracle.raster, oracle.texmlp, oracle.seq_ops; "
            "import importlib.util as u; s = u.spec_from_file_location('_p', %r); m = u.module_from_spec(s); s.loader.exec_module(m); "
            "import torch; t = m.MobileNetPerceptualLoss(use_gpu=False, seed=1); x = torch.rand(1, 3, 32, 32); t(x, torch.rand(1, 3, 32, 32)); "
            "bad = [k for k in sys.modules if k == 'd3h' or k.startswith('d3h.')]; assert not bad, bad") % (
        ROOT, os.path.join(ROOT, 'd3human-code_amd', 'geometry', 'perceptual.py'))
    r = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, cwd='/tmp',
                       env={k: v for k, v in os.environ.items() if k != 'PYTHONPATH'})
    assert r.returncode == 0, r.stderr[-2000:]
