"""Oracle for the SDF query: positional encoding + 8-layer softplus MLP (torch CPU, fp32).

Follows geometry/embedding.py:21-38 (Embedding.forward), geometry/mlp.py:10-45 (MLP) and the sweep
in geometry/hmsdf.py:433-444.  Pinned by tests/golden/sdf_mlp.npz (reference MLP outputs + grads).
"""
import torch
import torch.nn.functional as F

N_FREQ, D_HIDDEN, N_HIDDEN, SKIP_IN = 6, 256, 6, (3,)


def embed(x, n_freq=N_FREQ):
    # embedding.py:33-38: out = [x]; for freq in 2^0..2^(N-1): out += [sin(freq*x), cos(freq*x)]
    out = [x]
    for k in range(n_freq):
        f = float(2 ** k)
        out += [torch.sin(f * x), torch.cos(f * x)]
    return torch.cat(out, -1)


def layer_names(n_hidden=N_HIDDEN):
    # mlp.py:13-31: Linear at net.0, net.2, ..., net.(2*n_hidden+2); Softplus(beta=100) at odd indices
    return [2 * i for i in range(n_hidden + 2)]


def mlp_forward(x, sd, n_freq=N_FREQ, n_hidden=N_HIDDEN, skip_in=SKIP_IN, prefix='net.'):
    """sd: state_dict-like {f'net.{i}.weight', f'net.{i}.bias'}.  Returns [N,1]."""
    emb = embed(x, n_freq)
    h = emb
    idx = layer_names(n_hidden)
    for li, i in enumerate(idx):
        w, b = sd[f'{prefix}{i}.weight'], sd[f'{prefix}{i}.bias']
        if li >= 1 and (li - 1) in skip_in:      # mlp.py:21-23,40-41: hidden layer i in skip_in takes cat([x, emb])
            h = torch.cat([h, emb], -1)
        h = F.linear(h, w, b)
        if li != len(idx) - 1:
            h = F.softplus(h, beta=100)          # threshold=20 default, as nn.Softplus(beta=100)
    return h


def sdf_sweep(verts, deform, max_disp, sd, chunk=100000):
    # hmsdf.py:433-444
    v = verts + max_disp * deform if deform is not None else verts
    outs = [mlp_forward(v[i:i + chunk], sd) for i in range(0, v.shape[0], chunk)]
    return v, torch.cat(outs, 0)
