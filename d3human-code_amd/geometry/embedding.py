"""NeRF positional encoding with the reference's interface (geometry/embedding.py:4-38).  The SDF path fuses this into the MLP
kernel (csrc/sdf_mlp.hip); this module form is the library path used by MLP.forward_reference / MLP_deform."""
import torch
from torch import nn


class Embedding(nn.Module):
    def __init__(self, in_channels, N_freqs, logscale=True):
        super().__init__()
        self.N_freqs, self.in_channels = N_freqs, in_channels
        self.funcs = [torch.sin, torch.cos]
        self.out_channels = in_channels * (len(self.funcs) * N_freqs + 1)
        self.freq_bands = 2 ** torch.linspace(0, N_freqs - 1, N_freqs) if logscale else torch.linspace(1, 2 ** (N_freqs - 1), N_freqs)

    def forward(self, x):
        out = [x]
        for f in self.freq_bands.tolist():
            out += [torch.sin(f * x), torch.cos(f * x)]
        return torch.cat(out, -1)
