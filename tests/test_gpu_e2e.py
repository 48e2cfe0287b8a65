"""-m gpu: one whole training tick (SDF sweep -> marching tets -> LBS -> rasterize -> losses -> backward) on the HIP path against the
reference's own tick_init (golden) and against the oracle chain, every loss term and every parameter gradient."""
import pytest

import e2e_cases as E

pytestmark = pytest.mark.gpu


def test_gpu_tick_init_matches_reference_golden(gpu):
    worst = E.check_tick_init_golden(gpu)
    print('tick_init golden, worst relative gradient error per tensor:', {k: f'{v:.1e}' for k, v in worst.items()})


def test_gpu_tick_init_default_path_vs_oracle_chain(gpu):
    """config-3 loss stack (mask + normal + SSIM + sdf_reg + eikonal), 2 frames, fused pixel losses + loss head"""
    for kw in (dict(n=14, res=80, frames=2, seed=0), dict(n=16, res=96, frames=2, seed=1, iteration=700), dict(n=12, res=64, frames=3, seed=2)):
        worst = E.check_tick_init_vs_oracle(gpu, **kw)
        print(kw, {k: f'{v:.1e}' for k, v in worst.items() if v > 1e-4})


def test_gpu_tick_init_mask_only_vs_oracle_chain(gpu):
    """config-2 loss set (mask loss only, 1 frame)"""
    E.check_tick_init_vs_oracle(gpu, n=16, res=96, frames=1, seed=3, loss_set='mask', ssim_weight=0.0, n_samples=500)
