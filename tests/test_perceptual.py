"""MobileNetV2 perceptual loss trunk: torchvision-compatible state_dict layout (so a real checkpoint drops in), loss semantics."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'd3human-code_amd'))

# shapes of torchvision.models.mobilenet_v2().features[0..7] (published architecture, width multiplier 1.0)
EXPECTED = {
    '0.0.weight': (32, 3, 3, 3), '0.1.weight': (32,), '0.1.running_var': (32,),
    '1.conv.0.0.weight': (32, 1, 3, 3), '1.conv.1.weight': (16, 32, 1, 1), '1.conv.2.bias': (16,),
    '2.conv.0.0.weight': (96, 16, 1, 1), '2.conv.1.0.weight': (96, 1, 3, 3), '2.conv.2.weight': (24, 96, 1, 1), '2.conv.3.running_mean': (24,),
    '3.conv.0.0.weight': (144, 24, 1, 1), '3.conv.2.weight': (24, 144, 1, 1),
    '4.conv.1.0.weight': (144, 1, 3, 3), '4.conv.2.weight': (32, 144, 1, 1),
    '6.conv.0.0.weight': (192, 32, 1, 1), '7.conv.2.weight': (64, 192, 1, 1), '7.conv.3.weight': (64,),
}


def test_state_dict_layout_and_loss():
    from geometry.perceptual import MobileNetPerceptualLoss
    m = MobileNetPerceptualLoss(use_gpu=False)
    sd = m.features.state_dict()
    for k, shp in EXPECTED.items():
        assert k in sd and tuple(sd[k].shape) == shp, k
    assert len(m.features) == 8 and not any(p.requires_grad for p in m.features.parameters())
    g = torch.Generator().manual_seed(0)
    x = torch.rand(1, 3, 64, 64, generator=g, requires_grad=True)
    y = torch.rand(1, 3, 64, 64, generator=g)
    assert float(m(y, y)) == 0.0
    l = m(x, y)
    assert float(l) > 0
    l.backward()
    assert torch.isfinite(x.grad).all() and x.grad.abs().max() > 0
    # strides: layer 2 -> /4, layer 4 -> /8, layer 7 -> /16 (what the reference's [2, 4, 7] picks up)
    h = x.detach()
    res = {}
    for i, layer in enumerate(m.features):
        h = layer(h)
        res[i] = h.shape[-1]
    assert (res[2], res[4], res[7]) == (16, 8, 4)
    # a torchvision-style checkpoint (keys prefixed 'features.') loads; a truncated one is rejected
    ck = {'features.' + k: v.clone() + 0.01 for k, v in sd.items()}
    ck['classifier.1.weight'] = torch.zeros(1000, 1280)
    m2 = MobileNetPerceptualLoss(use_gpu=False, weights=ck)
    assert m2.pretrained and torch.allclose(m2.features.state_dict()['0.0.weight'], sd['0.0.weight'] + 0.01)
    ck.pop('features.4.conv.2.weight')
    try:
        MobileNetPerceptualLoss(use_gpu=False, weights=ck)
        assert False
    except RuntimeError:
        pass
