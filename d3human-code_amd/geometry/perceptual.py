"""MobileNetV2-feature perceptual loss of the reference (geometry/hmsdf.py:137-159): L1 between the activations of
`torchvision.models.mobilenet_v2(pretrained=True).features` at layers [2, 4, 7] of the two images, averaged over the three layers.

torchvision is not installed here and pretrained weights are unobtainable offline, so the trunk (features[0..7] is all the loss
reads) is rebuilt with torchvision's exact module structure: its state_dict keys and shapes are those of
`mobilenet_v2().features` (`0.0.weight`, `0.1.running_mean`, `2.conv.0.0.weight`, `2.conv.2.weight`, `2.conv.3.bias`, ...), so
`load_weights()` accepts a torchvision checkpoint (`mobilenet_v2-*.pth`, keys prefixed `features.`) unchanged.  Without a checkpoint
the trunk keeps its seeded random initialisation and the loss is a random-feature perceptual loss: usable as a smoke / throughput
stand-in, NOT a reproduction of the reference's values -- tick_* therefore only uses it when FLAGS.normal_loss_fn is set explicitly
(hmsdf.py of this build falls back to the reference's own MSE + cosine formula otherwise).  Convolutions run through MIOpen."""
import torch
import torch.nn as nn


def _conv_bn_relu6(inp, oup, kernel=3, stride=1, groups=1):
    return nn.Sequential(nn.Conv2d(inp, oup, kernel, stride, (kernel - 1) // 2, groups=groups, bias=False), nn.BatchNorm2d(oup),
                         nn.ReLU6(inplace=False))


class InvertedResidual(nn.Module):
    """torchvision.models.mobilenetv2.InvertedResidual (same sub-module names: `.conv` Sequential)"""

    def __init__(self, inp, oup, stride, expand_ratio):
        super().__init__()
        hidden = int(round(inp * expand_ratio))
        self.use_res_connect = stride == 1 and inp == oup
        layers = []
        if expand_ratio != 1:
            layers.append(_conv_bn_relu6(inp, hidden, kernel=1))
        layers += [_conv_bn_relu6(hidden, hidden, stride=stride, groups=hidden), nn.Conv2d(hidden, oup, 1, 1, 0, bias=False), nn.BatchNorm2d(oup)]
        self.conv = nn.Sequential(*layers)

    def forward(self, x):
        return x + self.conv(x) if self.use_res_connect else self.conv(x)


def mobilenet_v2_features(n_layers=8):
    """features[0 .. n_layers-1] of torchvision's MobileNetV2 (width 1.0): stem 32, then (t, c, n, s) = (1,16,1,1), (6,24,2,2),
    (6,32,3,2), (6,64,4,2), ..."""
    cfg = [(1, 16, 1, 1), (6, 24, 2, 2), (6, 32, 3, 2), (6, 64, 4, 2), (6, 96, 3, 1), (6, 160, 3, 2), (6, 320, 1, 1)]
    feats = [_conv_bn_relu6(3, 32, stride=2)]
    inp = 32
    for t, c, n, s in cfg:
        for i in range(n):
            feats.append(InvertedResidual(inp, c, s if i == 0 else 1, t))
            inp = c
    return nn.ModuleList(feats[:n_layers])


class MobileNetPerceptualLoss(nn.Module):
    def __init__(self, layers=(2, 4, 7), use_gpu=True, weights=None, seed=0):
        super().__init__()
        self.layers = list(layers)
        g = torch.random.get_rng_state()
        torch.manual_seed(seed)
        self.features = mobilenet_v2_features(max(self.layers) + 1).eval()
        torch.random.set_rng_state(g)
        self.pretrained = False
        if weights is not None:
            self.load_weights(weights)
        for p in self.features.parameters():
            p.requires_grad = False
        self.criterion = nn.L1Loss()
        if use_gpu and torch.cuda.is_available():
            self.features = self.features.cuda()

    def load_weights(self, path_or_state):
        sd = torch.load(path_or_state, map_location='cpu') if isinstance(path_or_state, str) else path_or_state
        sd = {k[len('features.'):]: v for k, v in sd.items() if k.startswith('features.')} or sd
        own = self.features.state_dict()
        missing = [k for k in own if k not in sd and not k.endswith('num_batches_tracked')]
        if missing:
            raise RuntimeError(f'MobileNetPerceptualLoss: checkpoint lacks {len(missing)} tensors, e.g. {missing[:3]}')
        self.features.load_state_dict({k: sd[k] for k in own if k in sd}, strict=False)
        self.pretrained = True

    def train(self, mode=True):          # the trunk is frozen and stays in eval mode (BatchNorm uses its running statistics)
        super().train(mode)
        self.features.eval()
        return self

    def forward(self, x, y):
        loss = 0
        for i, layer in enumerate(self.features):
            x = layer(x)
            y = layer(y)
            if i in self.layers:
                loss = loss + self.criterion(x, y)
        return loss / 3
