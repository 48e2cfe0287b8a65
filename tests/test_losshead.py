"""d3h/losshead.py: the affine loss head equals the reference's scalar formulas (hmsdf.py:835-839,881,895-898; train.py:718), values
and gradients."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'd3human-code_amd'))
from d3h.losshead import AffineHead


def test_affine_head_matches_scalar_formulas():
    torch.manual_seed(0)
    sw = 0.2
    x = torch.rand(12, dtype=torch.float32).requires_grad_(True)
    rows = {'img_loss': ({1: 1.0, 2: 0.5, 3: 0.5}, 0.0), 'msk_loss': ({0: 100.0}, 0.0), 'normal_loss': ({4: 1.0, 5: -0.1}, 0.1),
            'ssim_loss': ({9: -sw}, sw), 'reg_loss': ({10: 1.0, 11: 1.0}, 0.0),
            'total': ({0: 100.0, 4: 1.0, 5: -0.1, 9: -sw, 10: 1.0, 11: 1.0}, 0.1 + sw)}
    h = AffineHead(rows, 12, 'cpu')(x)
    y = x.detach().clone().requires_grad_(True)
    ref = {'img_loss': y[1] + 0.5 * y[2] + 0.5 * y[3], 'msk_loss': 100 * y[0], 'normal_loss': y[4] + 0.1 * (1 - y[5]),
           'ssim_loss': sw * (1.0 - y[9]), 'reg_loss': y[10] + y[11]}
    ref['total'] = ref['reg_loss'] + ref['normal_loss'] + ref['msk_loss'] + ref['ssim_loss']
    for k in ref:
        assert abs(float(h[k].detach()) - float(ref[k].detach())) <= 2e-6 * max(1.0, abs(float(ref[k].detach()))), k
    h['total'].backward()
    ref['total'].backward()
    assert torch.allclose(x.grad, y.grad, rtol=1e-6, atol=1e-7)
    # gradients through several named outputs at once (train.py-style callers add the entries themselves)
    x.grad = None
    h2 = AffineHead(rows, 12, 'cpu')(x)
    (h2['reg_loss'] + h2['normal_loss'] + h2['msk_loss'] + h2['ssim_loss']).backward()
    assert torch.allclose(x.grad, y.grad, rtol=1e-6, atol=1e-7)


def test_closed_form_lr_schedule_equals_lambdalr():
    """d3h.scene._LambdaLR sets exactly the learning rates torch.optim.lr_scheduler.LambdaLR would (train.py:573-576)"""
    from d3h.scene import _LambdaLR
    f = lambda it: it / 300 if it < 300 else max(0.0, 10 ** (-(it - 300) * 0.0002))
    mk = lambda: torch.optim.Adam([{'params': [torch.nn.Parameter(torch.zeros(1))], 'lr': 0.03},
                                   {'params': [torch.nn.Parameter(torch.zeros(1))], 'lr': 3e-4}])
    a, b = mk(), mk()
    sa, sb = torch.optim.lr_scheduler.LambdaLR(a, lr_lambda=f), _LambdaLR(b, f)
    for k in range(650):
        assert [g['lr'] for g in a.param_groups] == [g['lr'] for g in b.param_groups], k
        a.step(); sa.step(); sb.step()


def test_loss_callable_is_recognised_as_image_loss(emul):
    """train.py:75-87 (createLoss) hands tick_* bare lambdas around ru.image_loss; render.renderutils.loss_spec recognises them by one probe call
    (so that the unmodified train.py gets the fused per-pixel pass), and nothing that is not exactly that call"""
    import torch
    from render import renderutils as ru
    dev = torch.device('cpu')
    variants = {'smape': ('smape', 'none'), 'mse': ('mse', 'none'), 'logl1': ('l1', 'log_srgb'), 'logl2': ('mse', 'log_srgb'), 'relmse': ('relmse', 'none')}
    for name, (loss, tm) in variants.items():
        fn = (lambda l, t: (lambda img, ref: ru.image_loss(img, ref, loss=l, tonemapper=t)))(loss, tm)         # as createLoss builds them
        assert ru.loss_spec(fn, dev) == (loss, tm), name
        assert ru.loss_spec(fn, dev) == (loss, tm)                                                               # cached
    declared = lambda a, b: a.sum()
    declared.d3h_spec = ('l1', 'log_srgb')
    assert ru.loss_spec(declared, dev) == ('l1', 'log_srgb')
    assert ru.loss_spec(lambda a, b: 2 * ru.image_loss(a, b, loss='l1'), dev) is None                            # result changed
    assert ru.loss_spec(lambda a, b: ru.image_loss(a * 1.0, b, loss='l1'), dev) is None                          # input changed
    assert ru.loss_spec(lambda a, b: ru.image_loss(a, b) + ru.image_loss(a, b, loss='mse'), dev) is None         # two calls
    assert ru.loss_spec(lambda a, b: torch.nn.functional.l1_loss(a, b), dev) is None                             # not image_loss at all
    assert ru.loss_spec(lambda a, b: (_ for _ in ()).throw(RuntimeError('boom')), dev) is None                   # raises: left alone
    assert ru._IMAGE_LOSS_PROBE is None

    class Holder:                 # a bound method is a fresh object on every access: recognised once, then served from the cache
        def loss(self, a, b):
            return ru.image_loss(a, b, loss='l1', tonemapper='log_srgb')
    h = Holder()
    assert ru.loss_spec(h.loss, dev) == ('l1', 'log_srgb')
    n0 = len(ru._SPEC_CACHE)
    for _ in range(5):
        assert ru.loss_spec(h.loss, dev) == ('l1', 'log_srgb')
    assert len(ru._SPEC_CACHE) == n0
