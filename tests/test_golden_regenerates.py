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
    # parity_state_sdf.npz is not a reference output: a CPU-fitted SDF network, the fixed state of the whole-tick parity tests at BASELINE
    # sizes (tools/gen_parity_state.py, ~20 min of CPU; checked by test_parity_state_fixture_is_a_fitted_body below)
    committed = sorted(f for f in os.listdir(GOLD) if f.endswith('.npz') and f != 'parity_state_sdf.npz')
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
    """oracle/ is the checker: importing it (its own MobileNet-shaped trunk included) must pull in neither d3h nor any product module"""
    code = ("import sys; sys.path.insert(0, %r); import oracle.tick, oracle.parity, oracle.render, oracle.raster, oracle.texmlp, oracle.seq_ops, oracle.perceptual as m; "
            "import torch; t = m.MobileNetPerceptualLoss(use_gpu=False, seed=1); x = torch.rand(1, 3, 32, 32); t(x, torch.rand(1, 3, 32, 32)); "
            "bad = [k for k in sys.modules if k == 'd3h' or k.startswith('d3h.') or k.startswith('geometry') or k.startswith('render')]; assert not bad, bad") % (
        ROOT,)
    r = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, cwd='/tmp',
                       env={k: v for k, v in os.environ.items() if k != 'PYTHONPATH'})
    assert r.returncode == 0, r.stderr[-2000:]


def test_parity_state_fixture_is_a_fitted_body():
    """tests/golden/parity_state_sdf.npz (tools/gen_parity_state.py): the reference's network shape, finite, and -- through the oracle's
    restatement of the network -- the analytic capsule humanoid to the rmse the generator recorded"""
    import torch
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'd3human-code_amd'))
    from oracle import sdf_mlp as O
    from d3h import synth
    g = np.load(os.path.join(GOLD, 'parity_state_sdf.npz'))
    want = {0: (256, 39), 2: (256, 256), 4: (256, 256), 6: (256, 256), 8: (256, 295), 10: (256, 256), 12: (256, 256), 14: (1, 256)}
    sd = {}
    for i, shp in want.items():
        assert g[f'net.{i}.weight'].shape == shp and g[f'net.{i}.bias'].shape == (shp[0],) and g[f'net.{i}.weight'].dtype == np.float32
        assert np.isfinite(g[f'net.{i}.weight']).all()
        sd[f'net.{i}.weight'], sd[f'net.{i}.bias'] = torch.from_numpy(g[f'net.{i}.weight']), torch.from_numpy(g[f'net.{i}.bias'])
    v, _ = synth.kuhn_grid(24)
    v = torch.from_numpy(v)
    with torch.no_grad():
        err = O.mlp_forward(v, sd).reshape(-1) - synth.body_sdf(v)
    assert float(err.pow(2).mean().sqrt()) < 2.0 * float(g['fit_rmse']) + 1e-3
