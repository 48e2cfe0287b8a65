"""Shim under the third-party name so `import tinycudann as tcnn` (render/mlptexture.py:11) resolves to csrc/texmlp.hip.
Only what the reference uses: tcnn.Encoding(3, HashGrid cfg) with .params / .n_output_dims, and free_temporary_memory()."""
import torch

from d3h import texmlp as _T


class Encoding(torch.nn.Module):
    def __init__(self, n_input_dims, encoding_config, dtype=None, seed=1337):
        super().__init__()
        c = encoding_config
        ok = (n_input_dims == 3 and c.get('otype') == 'HashGrid' and c.get('n_levels') == 5 and c.get('n_features_per_level') == 2
              and c.get('base_resolution') == 16 and abs(c.get('per_level_scale') - _T.PER_LEVEL_SCALE) < 1e-9
              and c.get('log2_hashmap_size', 21) >= 19)
        if not ok:
            raise NotImplementedError(f'd3h tinycudann shim: only the HashGrid of render/mlptexture.py:68-75 is built (got {c})')
        self.n_input_dims = 3
        self.n_output_dims = _T.ENC_DIMS
        dev = 'cuda' if torch.cuda.is_available() else 'cpu'
        g = torch.Generator().manual_seed(seed)
        n = _T.grid_param_count()
        # tcnn initialises grid features U(-1e-4, 1e-4)
        self.params = torch.nn.Parameter(((torch.rand(n, generator=g) * 2 - 1) * 1e-4).to(dev))

    def forward(self, x):
        return _T.grid_encode(x, self.params)


def free_temporary_memory():
    pass
