"""The bf16 x 3 operand split of the SDF network's GEMMs (csrc/sdf_mlp_x3.h): pack layout, fp32-level agreement of every sweep with the
exact-f32 MFMA kernels and with the reference's own outputs (golden), on the host emulation of the kernel sources; GPU twins at full
size in test_gpu_fullsize.py."""
import numpy as np
import pytest
import torch

from conftest import golden
from parity_cases import T, sd_from_golden


def _bf16_planes_to_f32(dw):
    """dword array of bf16 pairs -> (low, high) fp32 values"""
    lo = (dw.astype(np.uint32) << np.uint32(16)).view(np.float32)
    hi = (dw.astype(np.uint32) & np.uint32(0xffff0000)).view(np.float32)
    return lo, hi


def test_emul_pack3_planes_sum_to_the_weights(emul):
    """wpack3 [rbg][kb][part 3][lane 64][4 dwords]: the three planes of every hidden-layer weight add up to the fp32 weight to 2^-23
    relative (3 x 8 significant bits), at the (row, k-step) the MFMA operand layout prescribes; the fp32 tail is the biases and the head"""
    from d3h import sdf_mlp
    from d3h import _lib as L
    g = golden('sdf_mlp.npz')
    sd = sd_from_golden(g, 'cpu')
    wp3 = sdf_mlp.pack_weights3(sd).numpy().view(np.uint32)
    assert wp3.shape[0] == L.lib().d3h_sdf_mlp_wpack3_dwords()
    off_l1 = 2 * 8 * 2 * 3 * 256                      # two layer-0 chunks of [rbl 8][kb 2][part 3][256 dwords]
    W = g['sd.net.2.weight']                          # layer 1 (net.2): 8 chunks of [rbl 2][kb 8][part 3][lane 64][4]
    blk = wp3[off_l1:off_l1 + 16 * 8 * 3 * 256].reshape(16, 8, 3, 64, 4)
    rec = np.zeros((256, 256), np.float64)
    for part in range(3):
        lo, hi = _bf16_planes_to_f32(blk[:, :, part])           # [rbg 16][kb 8][lane 64][d 4]
        for d in range(4):
            for e, v in ((0, lo[..., d]), (1, hi[..., d])):
                s = 2 * d + e
                for lane in range(64):
                    i, q = lane & 15, lane >> 4
                    feat = 32 * np.arange(8) + 16 * (s >> 2) + 4 * q + (s & 3)          # x3_feature(kb, q, s)
                    rec[16 * np.arange(16)[:, None] + i, feat[None, :]] += v[:, :, lane]
    err = np.abs(rec - W.astype(np.float64)).max() / np.abs(W).max()
    assert err < 2.0 ** -22, err
    tail = wp3[-L.lib().d3h_sdf_mlp_wpack_floats() + (L.lib().d3h_sdf_mlp_wpack_floats() - 2052):].view(np.float32)
    assert np.array_equal(tail[:256], g['sd.net.0.bias'])
    assert np.array_equal(tail[7 * 256:8 * 256], g['sd.net.14.weight'].reshape(-1))


def test_emul_x3_forward_matches_exact_f32_and_reference(emul):
    from d3h import sdf_mlp
    g = golden('sdf_mlp.npz')
    sd = sd_from_golden(g, 'cpu')
    n = 150                                            # ragged: 9 full wave tiles + 6 points
    x = T(g['x'], 'cpu')[:n].contiguous()
    ref = g['sdf'].reshape(-1)[:n]
    wp, wp3 = sdf_mlp.pack_weights(sd), sdf_mlp.pack_weights3(sd)
    o1, a1, _ = sdf_mlp.forward(x, wp, save=True)
    o3, a3, _ = sdf_mlp.forward(x, wp, save=True, wp3=wp3)
    e1, e3 = np.abs(o1.numpy() - ref).max(), np.abs(o3.numpy() - ref).max()
    assert e3 < 2e-7 and e3 <= 2 * e1 + 3e-8, (e1, e3)                    # no further from the reference's output than the f32 MFMA path
    assert np.array_equal(o3.numpy() > 0, ref > 0)
    nreal = (n + 15) // 16 * 7 * 4096
    assert (a3[:nreal] - a1[:nreal]).abs().max() <= 2e-6 * a1[:nreal].abs().max()     # the saved activations feed the same backward kernels
    # deformed sweep (hmsdf.py:433) and the no-save form
    dfm = torch.randn(n, 3) * 0.1
    assert torch.allclose(sdf_mlp.forward(x, wp, deform=dfm, disp=0.05, wp3=wp3), sdf_mlp.forward(x, wp, deform=dfm, disp=0.05), atol=1e-7)


@pytest.mark.parametrize('x3', [True, False])
def test_emul_backward_and_eikonal_on_both_arithmetics(emul, x3, monkeypatch):
    """the reference-autograd goldens of the first-order backward and the torch double-backward check of the eikonal term, with the bf16 x 3
    sweeps (default) and with the exact-f32 MFMA kernels (D3H_SDF_X3=0)"""
    from d3h import sdf_mlp
    import parity_cases as P
    monkeypatch.setattr(sdf_mlp, 'X3', x3)
    P.check_sdf_mlp_backward('cpu', n=40, tol=5e-5)
    P.check_sdf_mlp_backward('cpu', n=150, tol=5e-5, sparse_gout=True)
    P.check_sdf_mlp_eikonal('cpu', n=40, tol=2e-4)


def test_emul_no_save_sweep_with_recompute_equals_the_stored_activation_path(emul):
    """round 5: the training sweep runs without the 1.88 GB activation store; the sparse backward recomputes the activations of the tiles it
    visits with the same kernel.  Same sdf, same d(x), same parameter gradients as the path that stores them (D3H_SDF_RECOMPUTE=0), dense and
    sparse upstream gradients, with and without the deformation input"""
    from d3h import sdf_mlp
    g = golden('sdf_mlp.npz')
    sd = sd_from_golden(g, 'cpu')
    n = 333                                                   # not a multiple of 16 or 128
    gen = torch.Generator().manual_seed(5)
    x0 = T(g['x'][:n], 'cpu')
    deform0 = torch.randn(n, 3, generator=gen) * 0.01
    res = {}
    for rec in (False, True):
        for sparse in (False, True):
            sdf_mlp.RECOMPUTE = rec
            ps = [p.clone().requires_grad_(True) for p in sd.values()]
            x, deform = x0.clone().requires_grad_(True), deform0.clone().requires_grad_(True)
            out = sdf_mlp.sdf_query(x, ps, deform=deform, disp=0.5)
            go = torch.randn(n, 1, generator=torch.Generator().manual_seed(7))
            if sparse:
                go[torch.rand(n, 1, generator=torch.Generator().manual_seed(8)) < 0.9] = 0.0
            (out * go).sum().backward()
            res[(rec, sparse)] = (out.detach().clone(), x.grad.clone(), deform.grad.clone(), [p.grad.clone() for p in ps])
    sdf_mlp.RECOMPUTE = True
    for sparse in (False, True):
        a, b = res[(False, sparse)], res[(True, sparse)]
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2]), sparse
        for ga, gb in zip(a[3], b[3]):
            assert (ga - gb).abs().max() <= 1e-6 * max(1.0, float(ga.abs().max()))


# ---- "h2": two fp16 planes, three products (round 6) -------------------------------------------------------------------------------------
def test_emul_pack_h2_planes_sum_to_the_weights(emul):
    """wpackh2 [rbg][kb][part 2][lane 64][4 dwords]: value plane + 2^-11 x residual plane reproduce every hidden-layer weight to 2^-22 relative
    (2 x 11 significant bits) at the (row, k-step) the MFMA operand layout prescribes; a weight outside the fp16 working range is poisoned"""
    from d3h import sdf_mlp
    from d3h import _lib as L
    g = golden('sdf_mlp.npz')
    sd = sd_from_golden(g, 'cpu')
    wp = sdf_mlp.pack_weights_h2(sd)
    assert wp.d3h_planes == 2
    wp = wp.numpy().view(np.uint32)
    assert wp.shape[0] == L.lib().d3h_sdf_mlp_wpackh2_dwords()
    off_l1 = 2 * 8 * 2 * 2 * 256
    W = g['sd.net.2.weight']
    blk = wp[off_l1:off_l1 + 16 * 8 * 2 * 256].reshape(16, 8, 2, 64, 4)
    rec = np.zeros((256, 256), np.float64)
    for part, scale in ((0, 1.0), (1, 2.0 ** -11)):
        dw = blk[:, :, part]
        lo = (dw & np.uint32(0xffff)).astype(np.uint16).view(np.float16).astype(np.float64)
        hi = (dw >> np.uint32(16)).astype(np.uint16).view(np.float16).astype(np.float64)
        for d in range(4):
            for e, v in ((0, lo[..., d]), (1, hi[..., d])):
                s = 2 * d + e
                for lane in range(64):
                    i, q = lane & 15, lane >> 4
                    feat = 32 * np.arange(8) + 16 * (s >> 2) + 4 * q + (s & 3)
                    rec[16 * np.arange(16)[:, None] + i, feat[None, :]] += scale * v[:, :, lane]
    rel = np.abs(rec - W.astype(np.float64)) / np.maximum(np.abs(W), 1e-3)
    assert rel.max() < 2.0 ** -21, rel.max()
    bad = {k: v.clone() for k, v in sd.items()}
    bad['net.2.weight'][3, 5] = 7e4
    out = sdf_mlp.forward(T(g['x'], 'cpu')[:16].contiguous(), None, wp3=sdf_mlp.pack_weights_h2(bad))
    assert torch.isnan(out).all()                    # never a silently overflowed sweep


def test_emul_h2_forward_matches_exact_f32_and_reference(emul):
    from d3h import sdf_mlp
    g = golden('sdf_mlp.npz')
    sd = sd_from_golden(g, 'cpu')
    n = 150
    x = T(g['x'], 'cpu')[:n].contiguous()
    ref = g['sdf'].reshape(-1)[:n]
    wp, wph = sdf_mlp.pack_weights(sd), sdf_mlp.pack_weights_h2(sd)
    o1, a1, _ = sdf_mlp.forward(x, wp, save=True)
    o2, a2, _ = sdf_mlp.forward(x, wp, save=True, wp3=wph)
    e1, e2 = np.abs(o1.numpy() - ref).max(), np.abs(o2.numpy() - ref).max()
    assert e2 < 2e-7 and e2 <= 2 * e1 + 3e-8, (e1, e2)
    assert np.array_equal(o2.numpy() > 0, ref > 0)
    nreal = (n + 15) // 16 * 7 * 4096
    assert (a2[:nreal] - a1[:nreal]).abs().max() <= 2e-6 * a1[:nreal].abs().max()
    dfm = torch.randn(n, 3) * 0.1
    assert torch.allclose(sdf_mlp.forward(x, wp, deform=dfm, disp=0.05, wp3=wph), sdf_mlp.forward(x, wp, deform=dfm, disp=0.05), atol=1e-7)


def test_emul_h2_on_the_fitted_network(emul):
    """the CPU-fitted body network of the parity tests (weights up to O(1), |sdf| up to 1.6): h2 sweep vs the oracle's plain-torch evaluation"""
    import os
    from d3h import sdf_mlp
    from oracle import sdf_mlp as O
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'parity_state_sdf.npz'))
    sd = {k: torch.from_numpy(g[k]) for k in g.files if k.startswith('net.')}
    x = (torch.rand(96, 3, generator=torch.Generator().manual_seed(11)) * 2.4 - 1.2).contiguous()
    ref64 = O.mlp_forward(x.double(), {k: v.double() for k, v in sd.items()}).reshape(-1)
    ref32 = O.mlp_forward(x, sd).reshape(-1)
    out = sdf_mlp.forward(x, None, wp3=sdf_mlp.pack_weights_h2(sd))
    e2, e1 = float((out.double() - ref64).abs().max()), float((ref32.double() - ref64).abs().max())
    assert e2 <= 2 * e1 + 1e-7, (e2, e1)


@pytest.mark.parametrize('gscale', [1.0, 3e-7, 2e4])
def test_emul_h2_data_backward_is_scale_free(emul, gscale, monkeypatch):
    """the fp16 x 2 data-backward works on gradients scaled by a per-launch power of two (csrc/sdf_mlp_x3.h: h2_grad_scale): the same relative
    accuracy whatever the magnitude of the upstream gradient -- 3e-7 (a mean over 10^6 pixels) and 2e4 alike -- against the bf16 x 3 sweep,
    sparse and dense upstream gradients, with a deformation input"""
    from d3h import sdf_mlp
    g = golden('sdf_mlp.npz')
    sd = sd_from_golden(g, 'cpu')
    n = 200
    x0 = T(g['x'][:n], 'cpu')
    deform0 = torch.randn(n, 3, generator=torch.Generator().manual_seed(5)) * 0.01
    res = {}
    for h2b in (True, False):
        monkeypatch.setattr(sdf_mlp, 'H2_BWD', h2b)
        for sparse in (False, True):
            ps = [p.clone().requires_grad_(True) for p in sd.values()]
            x, deform = x0.clone().requires_grad_(True), deform0.clone().requires_grad_(True)
            pk = sdf_mlp.PackedWeights(ps)
            assert getattr(pk.wpt3, 'd3h_planes', 3) == (2 if h2b else 3)
            out = sdf_mlp.sdf_query(x, ps, deform=deform, disp=0.5, pack=pk)
            go = torch.randn(n, 1, generator=torch.Generator().manual_seed(7)) * gscale
            go[::9] *= 1e-4                                           # a wide spread of magnitudes inside one launch
            if sparse:
                go[torch.rand(n, 1, generator=torch.Generator().manual_seed(8)) < 0.8] = 0.0
            (out * go).sum().backward()
            res[(h2b, sparse)] = (x.grad.clone(), deform.grad.clone(), [p.grad.clone() for p in ps])
    for sparse in (False, True):
        a, b = res[(True, sparse)], res[(False, sparse)]
        for ga, gb in zip([a[0], a[1]] + a[2], [b[0], b[1]] + b[2]):
            assert torch.isfinite(ga).all()
            assert (ga - gb).abs().max() <= 2e-5 * gb.abs().max() + 1e-30, (sparse, float((ga - gb).abs().max()), float(gb.abs().max()))
        # per point: a point whose upstream gradient is 1e-4 of the launch's largest keeps its own relative accuracy
        rows = (b[0].abs().max(1).values > 0)
        rel = ((a[0] - b[0]).abs().max(1).values / b[0].abs().max(1).values.clamp(min=1e-38))[rows]
        assert float(rel.max()) <= 1e-3, float(rel.max())
