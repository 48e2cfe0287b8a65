"""LPIPS (d3human-code_amd/lpips) against the reference's vendored module (tests/golden/lpips.npz: third_parties/lpips/lpips.py run on a
seeded random trunk with its calibrated linear layers): values per layer, total, gradient; state_dict layout."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'd3human-code_amd'))


def _build(net, g, dev='cpu'):
    import lpips
    m = lpips.LPIPS(net=net, pretrained=False, trunk_seed=int(g['trunk_seed']))
    m.load_state_dict({f'lin{k}.model.1.weight': torch.from_numpy(g[f'{net}.lin{k}']) for k in range(5)}, strict=False)
    return m.to(dev)


def check_lpips_golden(dev, tol=2e-5):
    g = np.load(os.path.join(ROOT, 'tests', 'golden', 'lpips.npz'))
    a, b = torch.from_numpy(g['in0']).to(dev), torch.from_numpy(g['in1']).to(dev)
    for net in ('alex', 'vgg'):
        m = _build(net, g, dev)
        x = a.clone().requires_grad_(True)
        val, per = m(x, b, retPerLayer=True)
        ref = torch.from_numpy(g[f'{net}.val'])
        assert val.shape == ref.shape == (2, 1, 1, 1)
        assert (val.detach().cpu() - ref).abs().max() < tol * float(ref.abs().max()) + 1e-7, (net, val.reshape(-1).tolist(), ref.reshape(-1).tolist())
        for k, r in enumerate(per):
            rr = torch.from_numpy(g[f'{net}.layer{k}'])
            assert (r.detach().cpu() - rr).abs().max() < tol * float(rr.abs().max()) + 1e-7, (net, k)
        val.sum().backward()
        gr = torch.from_numpy(g[f'{net}.d_in0'])
        assert (x.grad.cpu() - gr).abs().max() < 50 * tol * float(gr.abs().max()), (net, float((x.grad.cpu() - gr).abs().max()), float(gr.abs().max()))
        assert float(m(b, b).abs().max()) < 1e-10                     # identical images: distance 0 (to the convolution library's run-to-run rounding)
        # the reference side handed in as cached features (what tick_split does with its constant targets): the same value and gradient
        y = a.clone().requires_grad_(True)
        val2 = m(y, None, ref_features=m.reference_features(b))
        assert (val2.detach() - val.detach()).abs().max() <= 1e-6 * float(val.detach().abs().max())
        val2.sum().backward()
        assert (y.grad - x.grad).abs().max() <= 1e-5 * float(x.grad.abs().max())


def test_lpips_matches_reference_golden_cpu():
    check_lpips_golden('cpu')


def check_fused_head(dev, tol=2e-5):
    """the one-pass head of a layer (csrc/lpips_head.hip through d3h.imgops.lpips_head: used when the metric is frozen) against the torch
    formulation of third_parties/lpips/lpips.py:112-134 -- per-layer values, total, and the gradient w.r.t. the first image -- on the
    golden's inputs; plus ragged sizes (H*W not a multiple of 256, a batch of 3)"""
    import lpips
    g = np.load(os.path.join(ROOT, 'tests', 'golden', 'lpips.npz'))
    a, b = torch.from_numpy(g['in0']).to(dev), torch.from_numpy(g['in1']).to(dev)
    for net in ('alex', 'vgg'):
        m = _build(net, g, dev)
        for p in m.parameters():
            p.requires_grad_(False)
        out = {}
        for fused in (False, True):
            lpips.FUSED_HEAD = fused
            x = a.clone().requires_grad_(True)
            val, per = m(x, b, retPerLayer=True)
            val.sum().backward()
            out[fused] = (val.detach().cpu(), [r.detach().cpu() for r in per], x.grad.cpu())
        lpips.FUSED_HEAD = True
        ref = torch.from_numpy(g[f'{net}.val'])
        assert (out[True][0] - ref).abs().max() < tol * float(ref.abs().max()) + 1e-7
        for r0, r1 in zip(out[False][1], out[True][1]):
            assert (r0 - r1).abs().max() <= tol * float(r0.abs().max()) + 1e-8
        assert (out[True][2] - out[False][2]).abs().max() <= 20 * tol * float(out[False][2].abs().max())
    # channels-last feature maps (what the trunk produces for an NHWC-strided image, e.g. tick_split's `.permute(0, 3, 1, 2)` view): read
    # in place by the 16-lanes-per-pixel kernels
    m = _build('alex', g, dev)
    for p in m.parameters():
        p.requires_grad_(False)
    x = a.clone().contiguous(memory_format=torch.channels_last).requires_grad_(True)
    v_cl = m(x, b.contiguous(memory_format=torch.channels_last))
    v_cl.sum().backward()
    assert (v_cl.detach().cpu() - torch.from_numpy(g['alex.val'])).abs().max() < tol * float(np.abs(g['alex.val']).max()) + 1e-7
    gr = torch.from_numpy(g['alex.d_in0'])
    assert (x.grad.cpu() - gr).abs().max() < 50 * tol * float(gr.abs().max())
    from d3h import imgops as I
    gen = torch.Generator().manual_seed(2)
    for C, cl in ((128, True), (64, True)):
        f0 = torch.rand(2, C, 7, 13, generator=gen).to(dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
        n1 = lpips.normalize_tensor(torch.rand(2, C, 7, 13, generator=gen)).to(dev).contiguous(memory_format=torch.channels_last)
        w = torch.randn(1, C, 1, 1, generator=gen).to(dev)
        wgt = torch.tensor([0.7, -1.3], device=dev)
        out = I.lpips_head(f0, n1, w)
        (out * wgt).sum().backward()
        got, f0.grad = f0.grad.clone(), None
        refv = torch.nn.functional.conv2d((lpips.normalize_tensor(f0) - n1) ** 2, w).mean([2, 3]).reshape(-1)
        (refv * wgt).sum().backward()
        assert (out.detach() - refv.detach()).abs().max() <= 1e-5 * float(refv.detach().abs().max()), C
        assert (got - f0.grad).abs().max() <= 1e-4 * float(f0.grad.abs().max()), C
    f0 = torch.rand(3, 37, 9, 31, generator=gen).to(dev).requires_grad_(True)
    n1 = lpips.normalize_tensor(torch.rand(3, 37, 9, 31, generator=gen)).to(dev)
    w = torch.randn(1, 37, 1, 1, generator=gen).to(dev)
    wgt = torch.tensor([0.3, -1.1, 2.0], device=dev)
    (I.lpips_head(f0, n1, w) * wgt).sum().backward()
    got, f0.grad = f0.grad.clone(), None
    d = (lpips.normalize_tensor(f0) - n1) ** 2
    refv = torch.nn.functional.conv2d(d, w).mean([2, 3]).reshape(-1)
    (refv * wgt).sum().backward()
    assert (I.lpips_head(f0.detach(), n1, w) - refv.detach()).abs().max() <= 1e-5 * float(refv.detach().abs().max())
    assert (got - f0.grad).abs().max() <= 1e-4 * float(f0.grad.abs().max())


def test_emul_lpips_fused_head(emul):
    check_fused_head(emul)


def test_lpips_state_dict_layout_and_errors():
    import lpips
    m = lpips.LPIPS(net='alex', pretrained=False)
    keys = set(m.state_dict().keys())
    assert {'scaling_layer.shift', 'scaling_layer.scale', 'net.slice1.0.weight', 'net.slice2.3.bias', 'net.slice5.10.weight', 'lin0.model.1.weight',
            'lin4.model.1.weight', 'lins.2.model.1.weight'} <= keys
    v = lpips.LPIPS(net='vgg', pretrained=False)
    assert {'net.slice1.0.weight', 'net.slice1.2.weight', 'net.slice3.14.weight', 'net.slice5.28.bias'} <= set(v.state_dict().keys())
    assert not any(p.requires_grad for p in m.net.parameters())
    with pytest.raises(FileNotFoundError):
        lpips.LPIPS(net='alex')                                       # calibrated linear layers need the reference's weight file
    with pytest.raises(NotImplementedError):
        lpips.LPIPS(net='squeeze', pretrained=False)
    x = torch.rand(1, 3, 32, 32)
    s = lpips.LPIPS(net='alex', pretrained=False, spatial=True)(x, torch.rand(1, 3, 32, 32))
    assert s.shape == (1, 1, 32, 32)
    assert torch.allclose(lpips.normalize_tensor(torch.ones(1, 4, 2, 2)), torch.full((1, 4, 2, 2), 0.5), atol=1e-6)


@pytest.mark.gpu
def test_gpu_lpips_matches_reference_golden(gpu):
    check_lpips_golden('cuda', tol=5e-5)
    check_fused_head('cuda', tol=5e-5)


@pytest.mark.gpu
def test_gpu_mobilenet_perceptual_loss_matches_cpu_evaluation(gpu):
    """the MobileNetV2-feature normal loss of hmsdf.py:137-159 (seeded trunk): MIOpen evaluation on the GPU == the CPU evaluation"""
    from geometry.perceptual import MobileNetPerceptualLoss
    gen = torch.Generator().manual_seed(2)
    x, y = torch.rand(1, 3, 256, 256, generator=gen), torch.rand(1, 3, 256, 256, generator=gen)
    mc = MobileNetPerceptualLoss(use_gpu=False, seed=5)
    mg = MobileNetPerceptualLoss(use_gpu=True, seed=5)
    assert next(mg.features.parameters()).is_cuda
    xc, xg = x.clone().requires_grad_(True), x.clone().cuda().requires_grad_(True)
    lc, lg = mc(xc, y), mg(xg, y.cuda())
    assert abs(float(lc) - float(lg)) < 2e-4 * abs(float(lc))
    lc.backward(); lg.backward()
    assert (xg.grad.cpu() - xc.grad).abs().max() < 2e-3 * xc.grad.abs().max()
