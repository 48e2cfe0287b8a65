"""Oracle for the MobileNetV2-feature perceptual loss (TEST INFRASTRUCTURE): geometry/hmsdf.py:137-159 restated in plain torch.

The reference takes `torchvision.models.mobilenet_v2(pretrained=True).features` and averages the L1 distances of the activations after
feature layers 2, 4 and 7.  torchvision and its weights are absent offline, so the trunk is rebuilt here module by module with
torchvision's structure (stem conv-BN-ReLU6 32, then inverted residuals (t, c, n, s) = (1,16,1,1), (6,24,2,2), (6,32,3,2), (6,64,4,2), ...;
same state_dict keys) and a SEEDED default initialisation.  This file shares no code with the product's geometry/perceptual.py (which folds
the BatchNorms and runs fused activations): the two agree on the weights only because both construct the same torch modules in the same
order from the same seed -- which tests/test_perceptual.py checks.
"""
import torch
import torch.nn as nn

_CFG = [(1, 16, 1, 1), (6, 24, 2, 2), (6, 32, 3, 2), (6, 64, 4, 2), (6, 96, 3, 1), (6, 160, 3, 2), (6, 320, 1, 1)]


def _cbr(cin, cout, k=3, stride=1, groups=1):
    return nn.Sequential(nn.Conv2d(cin, cout, k, stride, (k - 1) // 2, groups=groups, bias=False), nn.BatchNorm2d(cout), nn.ReLU6())


class _Block(nn.Module):
    """torchvision.models.mobilenetv2.InvertedResidual"""

    def __init__(self, cin, cout, stride, t):
        super().__init__()
        mid = int(round(cin * t))
        self.res = stride == 1 and cin == cout
        seq = [] if t == 1 else [_cbr(cin, mid, k=1)]
        seq += [_cbr(mid, mid, stride=stride, groups=mid), nn.Conv2d(mid, cout, 1, 1, 0, bias=False), nn.BatchNorm2d(cout)]
        self.conv = nn.Sequential(*seq)

    def forward(self, x):
        return x + self.conv(x) if self.res else self.conv(x)


def trunk(n_layers, seed):
    """features[0 .. n_layers-1], eval mode, frozen; default torch initialisation under `seed` (the global generator is restored)"""
    keep = torch.random.get_rng_state()
    torch.manual_seed(seed)
    mods, cin = [_cbr(3, 32, stride=2)], 32
    for t, c, n, s in _CFG:
        for i in range(n):
            mods.append(_Block(cin, c, s if i == 0 else 1, t))
            cin = c
    torch.random.set_rng_state(keep)
    f = nn.ModuleList(mods[:n_layers]).eval()
    for p in f.parameters():
        p.requires_grad = False
    return f


class MobileNetPerceptualLoss(nn.Module):
    """hmsdf.py:137-159 (the constructor arguments of the build's class, so that either can be handed to oracle.tick.state_from_golden)"""

    def __init__(self, layers=(2, 4, 7), use_gpu=False, seed=0):
        super().__init__()
        self.layers = list(layers)
        self.features = trunk(max(self.layers) + 1, seed)
        self.criterion = nn.L1Loss()

    def forward(self, x, y):
        loss = 0
        for i, layer in enumerate(self.features):           # hmsdf.py:152-157
            x = layer(x)
            y = layer(y)
            if i in self.layers:
                loss = loss + self.criterion(x, y)
        return loss / 3
