"""Writes tests/golden/parity_state_sdf.npz: the SDF network of the whole-tick parity tests at BASELINE sizes (tests/test_gpu_fullsize.py),
fitted ON THE CPU in plain torch through the oracle's restatement of the network (oracle/sdf_mlp.py <- geometry/mlp.py:10-45,
geometry/embedding.py:21-38) to the analytic capsule-humanoid SDF -- the pre-fit of hmsdf.py:254-271 (Adam 1e-3, MSE against the
template's SDF on the grid vertices), on seeded mini-batches of the tet-res-128 grid.

Why a fixture: until round 5 those tests built their state with 300 pre-fit steps + 5 optimiser steps ON THE GPU, whose backward kernels
sum with float atomics -- every box compared a different scene, and the bars had been set from the states the builder happened to see
("What's weak" 1).  No GPU arithmetic touches this state: every box now compares the same network, the same mesh, the same pixels.

    python tools/gen_parity_state.py [steps] [batch]          (~10 min on 8 cores; the result is committed)
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'd3human-code_amd'))
from oracle import sdf_mlp as O                      # noqa: E402
from d3h import synth                                # noqa: E402  (host-side generators only: the Kuhn grid and the analytic SDF)


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 500
    batch = int(sys.argv[2]) if len(sys.argv) > 2 else 49152
    torch.manual_seed(0)
    # geometry/mlp.py:13-31 with the reference's configuration (n_freq 6, d_hidden 256, n_hidden 6, skip_in [3]); nn.Linear default init
    dims = [(39, 256), (256, 256), (256, 256), (256, 256), (256 + 39, 256), (256, 256), (256, 256), (256, 1)]
    sd = {}
    for i, (din, dout) in zip(O.layer_names(), dims):
        lin = torch.nn.Linear(din, dout)
        sd[f'net.{i}.weight'] = lin.weight.detach().clone().requires_grad_(True)
        sd[f'net.{i}.bias'] = lin.bias.detach().clone().requires_grad_(True)
    v, _ = synth.kuhn_grid(63)
    v = torch.from_numpy(v)
    gt = synth.body_sdf(v).reshape(-1, 1)
    opt = torch.optim.Adam(list(sd.values()), lr=1e-3)
    gen = torch.Generator().manual_seed(1)
    t0 = time.time()
    for it in range(steps):
        # half of every batch from the 12 % of the grid nearest to the surface: that is where the extracted mesh comes from
        near = torch.nonzero(gt.reshape(-1).abs() < 0.08).reshape(-1)
        i0 = torch.randint(0, v.shape[0], (batch // 2,), generator=gen)
        i1 = near[torch.randint(0, near.shape[0], (batch // 2,), generator=gen)]
        idx = torch.cat([i0, i1])
        loss = (O.mlp_forward(v[idx], sd) - gt[idx]).pow(2).mean()
        opt.zero_grad()
        loss.backward()
        opt.step()
        if it % 25 == 0 or it == steps - 1:
            print(f'step {it}: mse {float(loss):.3e}  ({time.time() - t0:.0f} s)', flush=True)
    with torch.no_grad():
        full = torch.cat([O.mlp_forward(v[i:i + 65536], sd) for i in range(0, v.shape[0], 65536)])
        err = (full - gt)
        print(f'full grid: rmse {float(err.pow(2).mean().sqrt()):.3e}, inside fraction {float((full < 0).float().mean()):.4f} '
              f'(analytic {float((gt < 0).float().mean()):.4f})')
    out = os.path.join(ROOT, 'tests', 'golden', 'parity_state_sdf.npz')
    np.savez_compressed(out, **{k: p.detach().numpy() for k, p in sd.items()}, fit_rmse=np.float32(err.pow(2).mean().sqrt()))
    print('wrote', out, os.path.getsize(out), 'bytes')


if __name__ == '__main__':
    main()
