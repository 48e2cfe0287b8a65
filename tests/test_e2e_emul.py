"""End-to-end tick parity on the host emulation of the kernels (kernel-logic debugging at toy size; the -m gpu versions in
tests/test_gpu_e2e.py are the parity tests proper) + the oracle chain against the reference golden (pure CPU)."""
import os

import numpy as np
import pytest
import torch

import e2e_cases as E
from conftest import golden


def test_oracle_chain_matches_reference_tick_init_golden():
    """oracle/tick.py == the reference's HmSDFTetsGeometry.tick_init (tests/golden/tick_init.npz): 6 loss terms, all gradients"""
    from geometry.perceptual import MobileNetPerceptualLoss
    from oracle import tick as OTK
    g = dict(golden('tick_init.npz'))
    st = OTK.state_from_golden(g, MobileNetPerceptualLoss)
    torch.manual_seed(int(g['draws_seed']))
    r = OTK.tick_init(st, buffers=None)
    for k in ('img_loss', 'msk_loss', 'eik_loss', 'sdf_reg_loss', 'reg_loss', 'normal_loss', 'total'):
        a, b = float(r[k]), float(g['loss.' + k])
        assert abs(a - b) <= 1e-5 * max(1e-3, abs(b)), (k, a, b)
    r['total'].backward()
    ref = {k[5:]: torch.from_numpy(g[k]) for k in g if k.startswith('grad.')}
    E._cmp_grads(E.oracle_grads(st), ref, 1e-3, 'oracle chain vs reference golden')


@pytest.mark.skipif(os.environ.get('D3H_SLOW_TESTS') != '1', reason='5 minutes on the host emulator (13^3 grid points and 2000 eikonal '
                    'samples through emulated MFMA sweeps); set D3H_SLOW_TESTS=1.  The -m gpu twin runs in seconds.')
def test_emul_tick_init_golden(emul):
    E.check_tick_init_golden(emul)


def test_emul_tick_init_vs_oracle_chain(emul):
    E.check_tick_init_vs_oracle(emul, n=6, res=32, frames=2, n_samples=96)
