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


class _BiasReLU6Fn(torch.autograd.Function):
    """clamp(x + b[None, :, None, None], 0, 6) on the device kernels of csrc/act_ops.hip: one pass each way (the library path is the bias
    add MIOpen issues after a convolution + the clamp forward, compare + multiply backward); no gradient for the frozen bias"""

    @staticmethod
    def forward(ctx, x, b):
        from d3h import _lib as L
        x = x.contiguous()
        N, C = x.shape[0], x.shape[1]
        HW = x.numel() // max(N * C, 1)
        y = torch.empty_like(x)
        L.check(L.lib().d3h_bias_relu6_fwd(L.ptr(x), L.ptr(b.contiguous()), L.i64(N), L.i32(C), L.i64(HW), L.ptr(y), L.stream()), 'bias_relu6_fwd')
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, g):
        from d3h import _lib as L
        y, = ctx.saved_tensors
        g = g.contiguous()
        gx = torch.empty_like(g)
        L.check(L.lib().d3h_relu6_bwd(L.ptr(y), L.ptr(g), L.i64(y.numel()), L.ptr(gx), L.stream()), 'relu6_bwd')
        return gx, None


def _fused_act(x):
    """device tensors (or the test emulator) take the fused bias + ReLU6 kernels; host tensors (the CPU goldens) the library ops.
    Host tensors never IMPORT d3h: this module also serves as the MobileNet-shaped trunk of the oracle tick (oracle/tick.py,
    tools/gen_golden.py), whose harness has no d3h on sys.path -- the emulator counts only if its fixture has loaded d3h._lib already"""
    if x.dtype != torch.float32:
        return False
    if x.is_cuda:
        return True
    import sys
    L = sys.modules.get('d3h._lib')
    return L is not None and L.emulated()


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

    # ---- the forward the ticks run: frozen trunk => BatchNorm folded into the convolutions, dense NCHW inputs, cached target features -----
    def _folded(self):
        """[(blocks of (weight, bias, stride, padding, groups, relu6), residual?)] per feature layer: eval-mode BatchNorm (running statistics,
        hmsdf.py:146-147 puts the trunk in eval()) folded into the preceding convolution -- w' = w * gamma / sqrt(var + eps), b' = beta -
        mean * gamma / sqrt(var + eps) -- so a conv-BN-ReLU6 triple is one convolution with bias + one clamp (the 42 BatchNorm launches and
        the elementwise passes around them were 2.4 ms of an 18.5 ms iteration at 1080 x 1080).  Rebuilt when the trunk's tensors change."""
        key = tuple((t.data_ptr(), t._version) for t in list(self.features.parameters()) + list(self.features.buffers()))
        hit = getattr(self, '_fold_cache', None)
        if hit is not None and hit[0] == key:
            return hit[1]

        def fold(conv, bn):
            with torch.no_grad():
                k = bn.weight / torch.sqrt(bn.running_var + bn.eps)
                return ((conv.weight * k[:, None, None, None]).contiguous(), (bn.bias - bn.running_mean * k).contiguous(),
                        conv.stride, conv.padding, conv.groups)
        out = []
        for layer in self.features:
            if isinstance(layer, InvertedResidual):
                mods, blocks = list(layer.conv), []
                j = 0
                while j < len(mods):
                    if isinstance(mods[j], nn.Sequential):                       # conv-BN-ReLU6
                        blocks.append(fold(mods[j][0], mods[j][1]) + (True,))
                        j += 1
                    else:                                                         # the linear 1 x 1 projection + its BatchNorm
                        blocks.append(fold(mods[j], mods[j + 1]) + (False,))
                        j += 2
                out.append((blocks, layer.use_res_connect))
            else:
                out.append(([fold(layer[0], layer[1]) + (True,)], False))
        self._fold_cache = (key, out)
        return out

    def _features(self, x):
        """activations of the layers the loss reads, through the folded trunk"""
        import torch.nn.functional as F
        feats = []
        for i, (blocks, res) in enumerate(self._folded()):
            h = x
            for w, b, stride, pad, groups, relu6 in blocks:
                if relu6 and _fused_act(h):
                    h = _BiasReLU6Fn.apply(F.conv2d(h, w, None, stride, pad, 1, groups), b)
                elif relu6:
                    h = F.hardtanh(F.conv2d(h, w, b, stride, pad, 1, groups), 0.0, 6.0)
                else:
                    h = F.conv2d(h, w, b, stride, pad, 1, groups)
            x = x + h if res else h
            if i in self.layers:
                feats.append(x)
        return feats

    def reference_features(self, y):
        """trunk activations of the TARGET image (a constant of the frame), for forward(x, None, ref_features=...)"""
        with torch.no_grad():
            return self._features(y.contiguous())

    def forward(self, x, y, ref_features=None):
        """mean over the three layers of L1(features(x), features(y)) (hmsdf.py:150-159).  The inputs arrive as NCHW VIEWS of NHWC images
        (`.permute(0, 3, 1, 2)` in tick_*): made dense NCHW here, because MIOpen's channels-last path has no tuned depthwise kernels and fell
        back to `naive_conv_*` (6.4 ms per iteration at 1080 x 1080).  The target side is a constant of the frame: its features are
        computed without a graph and cached per target TENSOR OBJECT (identity + version, as the LPIPS reference cache of tick_split)."""
        x = x.contiguous()
        if ref_features is None:
            if y.requires_grad:
                ref_features = self._features(y.contiguous())
            else:
                hit = getattr(self, '_ref_cache', None)
                fold_key = self._folded() is not None and self._fold_cache[0]
                if hit is None or hit[0] is not y or hit[1] != y._version or hit[2] != fold_key:
                    hit = self._ref_cache = (y, y._version, fold_key, self.reference_features(y))
                ref_features = hit[3]
        loss = 0
        for fx, fy in zip(self._features(x), ref_features):
            loss = loss + self.criterion(fx, fy)
        return loss / 3

    def forward_modules(self, x, y):
        """the module-by-module formulation (conv, BatchNorm, ReLU6 as separate launches): what forward() is checked against"""
        loss = 0
        for i, layer in enumerate(self.features):
            x = layer(x)
            y = layer(y)
            if i in self.layers:
                loss = loss + self.criterion(x, y)
        return loss / 3
