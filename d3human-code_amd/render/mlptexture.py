"""MLPTexture3D with the reference's interface (render/mlptexture.py:51-115): `.encoder.params`, `.net.net.{0,2,4}.weight`,
`.sample(texc, frame_id)`.  sample() runs the fused grid-encoding + MLP kernel (csrc/texmlp.hip)."""
import numpy as np
import torch
import tinycudann as tcnn

from d3h import texmlp as _T


class _MLP(torch.nn.Module):
    def __init__(self, cfg, loss_scale=1.0):
        super().__init__()
        self.loss_scale = loss_scale
        net = (torch.nn.Linear(cfg['n_input_dims'], cfg['n_neurons'], bias=False), torch.nn.ReLU())
        for _ in range(cfg['n_hidden_layers'] - 1):
            net = net + (torch.nn.Linear(cfg['n_neurons'], cfg['n_neurons'], bias=False), torch.nn.ReLU())
        net = net + (torch.nn.Linear(cfg['n_neurons'], cfg['n_output_dims'], bias=False),)
        self.net = torch.nn.Sequential(*net)
        for m in self.net:
            if isinstance(m, torch.nn.Linear):
                torch.nn.init.kaiming_uniform_(m.weight, nonlinearity='relu')     # mlptexture.py:36-41

    def forward(self, x):
        # library-GEMM path (only used when someone calls the sub-module directly; sample() uses the fused kernel)
        y = self.net(x.to(torch.float32))
        return y


class MLPTexture3D(torch.nn.Module):
    # mlptexture.py:94: hard-coded, sign-flipped box; AABB and frame_id are ignored by the reference -- kept literally
    BBOX = (0.6, 0.6, 0.2, -0.8, -1.2, -0.2)

    def __init__(self, AABB, channels=3, internal_dims=32, hidden=2, min_max=None, use_float16=False):
        super().__init__()
        self.channels, self.internal_dims, self.AABB, self.min_max, self.use_float16 = channels, internal_dims, AABB, min_max, use_float16
        per_level_scale = np.exp(np.log(4096 / 16) / (16 - 1))
        enc_cfg = {"otype": "HashGrid", "n_levels": 5, "n_features_per_level": 2, "log2_hashmap_size": 21, "base_resolution": 16,
                   "per_level_scale": per_level_scale}
        self.encoder = tcnn.Encoding(3, enc_cfg)
        self.net = _MLP({"n_input_dims": self.encoder.n_output_dims, "n_output_dims": channels, "n_hidden_layers": hidden,
                         "n_neurons": internal_dims}, 128.0)
        dev = self.encoder.params.device
        self.net.to(dev)
        if channels != 6 or internal_dims != 32 or hidden != 2:
            raise NotImplementedError('d3h MLPTexture3D: the fused kernel is built for the reference shape 10 -> 32 -> 32 -> 6')

    def _range_host(self):
        """host copy of the output range (HOST arguments of the C ABI).  Read back once per value: a `.cpu()` here is a stream
        synchronisation in the middle of every render otherwise, after which the rest of the forward is launch-bound."""
        mm = self.min_max
        lo, hi = mm[0], mm[1]                                           # a [2,C] tensor or a pair of tensors / lists
        key = tuple((t.data_ptr(), t._version) if torch.is_tensor(t) else None for t in (lo, hi))
        # the entry holds what owns the memory behind the key (the [2,C] tensor, or the two tensors of a pair), so that an address in the
        # key cannot be handed to another tensor while the entry lives
        own = (mm,) if torch.is_tensor(mm) else (lo, hi)
        hit = getattr(self, '_range_cache', None)
        if hit is None or hit[0] != key or None in key or len(hit[2]) != len(own) or any(a is not b for a, b in zip(hit[2], own)):
            host = lambda t: t.detach().float().cpu().tolist() if torch.is_tensor(t) else [float(v) for v in t]
            hit = self._range_cache = (key, (host(lo), host(hi)), own)
        return hit[1]

    def sample(self, texc, frame_id=None, mask=None):
        w = [self.net.net[i].weight for i in (0, 2, 4)]
        omin, omax = self._range_host()
        return _T.texture_mlp(texc, self.encoder.params, w[0], w[1], w[2], self.BBOX, omin, omax, mask=mask, in_grad_scale=self.net.loss_scale)

    def clamp_(self):
        pass

    def cleanup(self):
        tcnn.free_temporary_memory()
