"""LPIPS perceptual distance with the interface of the reference's vendored package (third_parties/lpips/lpips.py:19-145 `LPIPS`,
__init__.py:13-15 `normalize_tensor`; BASELINE config 5 names it in the full loss stack), built for the MI355X box:

 * the trunks (AlexNet / VGG-16 feature stacks, pretrained_networks.py:59-147) are rebuilt here with torchvision's exact layer indices, so
   the state_dict keys are the reference's (`net.slice3.7.weight`, `lin2.model.1.weight`, `scaling_layer.shift`, ...) and a checkpoint of the
   reference module loads with load_state_dict.  torchvision is not installed and ImageNet weights cannot be downloaded here: the trunk
   keeps a seeded random initialisation unless `trunk_weights` (a torchvision `alexnet` / `vgg16` state_dict or file) is given;
 * the calibrated linear layers are the vendored `weights/v0.1/{alex,vgg}.pth` of the reference (`model_path=`); they are data of the
   reference checkout and are not copied into this package;
 * convolutions / pooling run through MIOpen via torch.nn (this is a loss network outside the render-and-fit kernels; on-device, no host
   round trips).  SqueezeNet and the training utilities of the vendored package (trainer.py, Dist2LogitLayer) are not provided.
Pinned by tests/golden/lpips.npz: the reference's own LPIPS module on a seeded random trunk + the vendored linear weights."""
import torch
import torch.nn as nn


FUSED_HEAD = True          # module switch (tests compare both formulations)


def _emulated():
    try:
        from d3h import _lib
        return _lib.emulated()
    except Exception:
        return False


def normalize_tensor(in_feat, eps=1e-10):
    """unit-normalise the channel vector of every pixel (third_parties/lpips/__init__.py:13-15)"""
    return in_feat / (torch.sqrt(torch.sum(in_feat ** 2, dim=1, keepdim=True)) + eps)


def spatial_average(x, keepdim=True):
    return x.mean([2, 3], keepdim=keepdim)


def upsample(x, out_HW=(64, 64)):
    return nn.functional.interpolate(x, size=out_HW, mode='bilinear', align_corners=False)


# torchvision feature stacks as (index -> layer) tables; LPIPS taps the output of the last layer of every slice
_ALEX = [('conv', 3, 64, 11, 4, 2), ('relu',), ('pool', 3, 2), ('conv', 64, 192, 5, 1, 2), ('relu',), ('pool', 3, 2), ('conv', 192, 384, 3, 1, 1), ('relu',),
         ('conv', 384, 256, 3, 1, 1), ('relu',), ('conv', 256, 256, 3, 1, 1), ('relu',)]
_ALEX_SLICES = [(0, 2), (2, 5), (5, 8), (8, 10), (10, 12)]
_VGG_CFG = [64, 64, 'M', 128, 128, 'M', 256, 256, 256, 'M', 512, 512, 512, 'M', 512, 512, 512]
_VGG_SLICES = [(0, 4), (4, 9), (9, 16), (16, 23), (23, 30)]


def _vgg_layers():
    out, c = [], 3
    for v in _VGG_CFG:
        if v == 'M':
            out.append(('pool', 2, 2))
        else:
            out += [('conv', c, v, 3, 1, 1), ('relu',)]
            c = v
    return out


def _make(spec):
    if spec[0] == 'conv':
        return nn.Conv2d(spec[1], spec[2], spec[3], stride=spec[4], padding=spec[5])
    if spec[0] == 'relu':
        return nn.ReLU(inplace=False)
    return nn.MaxPool2d(kernel_size=spec[1], stride=spec[2])


class _Trunk(nn.Module):
    """slice1..slice5 Sequentials whose children carry the torchvision layer indices (pretrained_networks.py:59-147)"""

    def __init__(self, layers, slices, requires_grad=False, seed=0):
        super().__init__()
        state = torch.random.get_rng_state()
        torch.manual_seed(seed)
        self.N_slices = len(slices)
        for k, (a, b) in enumerate(slices):
            seq = nn.Sequential()
            for i in range(a, b):
                seq.add_module(str(i), _make(layers[i]))
            setattr(self, f'slice{k + 1}', seq)
        torch.random.set_rng_state(state)
        if not requires_grad:
            for p in self.parameters():
                p.requires_grad = False

    def forward(self, x):
        outs = []
        for k in range(self.N_slices):
            x = getattr(self, f'slice{k + 1}')(x)
            outs.append(x)
        return outs

    def load_torchvision(self, sd):
        """a torchvision alexnet / vgg16 state_dict (`features.<idx>.weight` ...) -> this trunk"""
        sd = torch.load(sd, map_location='cpu') if isinstance(sd, str) else sd
        own = self.state_dict()
        new = {}
        for k in own:
            idx_key = k.split('.', 1)[1]                       # 'slice3.7.weight' -> '7.weight'
            src = 'features.' + idx_key
            if src not in sd:
                raise RuntimeError(f'LPIPS trunk: checkpoint lacks {src}')
            new[k] = sd[src]
        self.load_state_dict(new)


class ScalingLayer(nn.Module):
    def __init__(self):
        super().__init__()
        self.register_buffer('shift', torch.tensor([-.030, -.088, -.188])[None, :, None, None])
        self.register_buffer('scale', torch.tensor([.458, .448, .450])[None, :, None, None])

    def forward(self, x):
        return (x - self.shift) / self.scale


class NetLinLayer(nn.Module):
    """1x1 convolution to one channel, no bias; `model.1.weight` when the Dropout of the reference's default sits at index 0"""

    def __init__(self, chn_in, chn_out=1, use_dropout=False):
        super().__init__()
        layers = [nn.Dropout()] if use_dropout else []
        layers += [nn.Conv2d(chn_in, chn_out, 1, stride=1, padding=0, bias=False)]
        self.model = nn.Sequential(*layers)

    def forward(self, x):
        return self.model(x)


class LPIPS(nn.Module):
    def __init__(self, pretrained=True, net='alex', version='0.1', lpips=True, spatial=False, pnet_rand=False, pnet_tune=False, use_dropout=True,
                 model_path=None, eval_mode=True, verbose=False, trunk_weights=None, trunk_seed=0):
        super().__init__()
        self.pnet_type, self.pnet_tune, self.pnet_rand, self.spatial, self.lpips, self.version = net, pnet_tune, pnet_rand, spatial, lpips, version
        self.scaling_layer = ScalingLayer()
        if net in ('vgg', 'vgg16'):
            self.chns = [64, 128, 256, 512, 512]
            self.net = _Trunk(_vgg_layers(), _VGG_SLICES, requires_grad=pnet_tune, seed=trunk_seed)
        elif net == 'alex':
            self.chns = [64, 192, 384, 256, 256]
            self.net = _Trunk(_ALEX, _ALEX_SLICES, requires_grad=pnet_tune, seed=trunk_seed)
        else:
            raise NotImplementedError(f"LPIPS trunk '{net}': only 'alex' and 'vgg' are built")
        self.L = len(self.chns)
        self.trunk_pretrained = False
        if trunk_weights is not None:
            self.net.load_torchvision(trunk_weights)
            self.trunk_pretrained = True
        if lpips:
            for k, c in enumerate(self.chns):
                setattr(self, f'lin{k}', NetLinLayer(c, use_dropout=use_dropout))
            self.lins = nn.ModuleList([getattr(self, f'lin{k}') for k in range(self.L)])
            if pretrained:
                if model_path is None:
                    raise FileNotFoundError("LPIPS(pretrained=True) needs model_path=<reference>/third_parties/lpips/weights/v0.1/%s.pth "
                                            "(the calibrated linear layers are data of the reference checkout)" % net)
                self.load_state_dict(torch.load(model_path, map_location='cpu'), strict=False)
        if eval_mode:
            self.eval()

    def _prepare(self, x, normalize):
        if normalize:                     # inputs in [0, 1] -> [-1, 1]
            x = 2 * x - 1
        return self.scaling_layer(x) if self.version == '0.1' else x

    @torch.no_grad()
    def reference_features(self, in1, normalize=True):
        """the unit-normalised trunk features of a reference image that takes no gradient (a training target: the same tensor in every
        iteration) -- hand them to forward(ref_features=...) and the trunk runs on the prediction only (extension; same value)"""
        return [normalize_tensor(f) for f in self.net(self._prepare(in1, normalize))]

    def prepare_constants(self, normalize=True):
        """(shift[3], scale[3]) of the input map ((2 x - 1) - shift) / scale that forward applies to in0 (normalize + ScalingLayer), as Python
        floats (read from the device once) -- for a producer that applies the map itself and calls forward(..., in0_prepared=True);
        None when this configuration maps differently"""
        if not normalize or self.version != '0.1':
            return None
        c = getattr(self, '_prep_consts', None)
        if c is None:
            c = self._prep_consts = (self.scaling_layer.shift.reshape(-1).tolist(), self.scaling_layer.scale.reshape(-1).tolist())
        return c

    def forward(self, in0, in1, retPerLayer=False, normalize=True, ref_features=None, in0_prepared=False):
        if not in0_prepared:              # (extension) in0 already is the trunk input: prepare_constants()
            in0 = self._prepare(in0, normalize)
        o0 = self.net(in0)
        n1 = ref_features if ref_features is not None else [normalize_tensor(f) for f in self.net(self._prepare(in1, normalize))]
        res = []
        for k in range(self.L):
            # fixed metric (the linear layers take no gradient), scalar output, device tensors: the whole head of the layer -- normalise,
            # squared difference, 1x1 convolution, spatial mean -- is one HIP pass (d3h.imgops.lpips_head); the torch formulation below
            # is what it is pinned against (tests/test_lpips.py)
            wk = self.lins[k].model[-1].weight if self.lpips else None
            if (FUSED_HEAD and self.lpips and not self.spatial and not wk.requires_grad and not n1[k].requires_grad and not self.training
                    and (o0[k].is_cuda or _emulated())):
                from d3h import imgops as _I
                res.append(_I.lpips_head(o0[k], n1[k], wk).view(-1, 1, 1, 1))
                continue
            d = (normalize_tensor(o0[k]) - n1[k]) ** 2
            d = self.lins[k](d) if self.lpips else d.sum(dim=1, keepdim=True)
            res.append(upsample(d, out_HW=in0.shape[2:]) if self.spatial else spatial_average(d, keepdim=True))
        val = res[0]
        for r in res[1:]:
            val = val + r
        return (val, res) if retPerLayer else val
