"""-m gpu: the hot path at BASELINE.json's full sizes (tet-res 128 = Kuhn n=63: 262 144 vertices / 1 500 282 tets; 1024^2 x 4 frames),
checked through the oracle where it finishes in seconds and through size-independent properties otherwise."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def grid63():
    from d3h import synth
    v, t = synth.kuhn_grid(63)
    return torch.from_numpy(v), torch.from_numpy(t)


def test_full_grid_marching_tets_bit_exact_vs_oracle(gpu, grid63):
    from d3h import mtets, synth
    from oracle import marching_tets as OMT
    v, t = grid63
    sdf = synth.body_sdf(v) + 0.003 * torch.sin(40 * v[:, 0]) * torch.cos(31 * v[:, 1])
    g = torch.Generator().manual_seed(1)
    msdf = (torch.rand(v.shape[0], generator=g) * 2 - 0.6)                     # mixed sign: every cut case occurs
    o = mtets.marching_tets(v.cuda(), sdf.cuda(), msdf.cuda(), t.cuda())
    ref = OMT.gshell_tets(v, sdf, msdf, t)
    assert o['faces'].shape[0] > 5000
    assert torch.equal(o['faces'].cpu(), ref['faces'])
    assert torch.equal(o['faces_wt'].cpu(), ref['faces_watertight'])
    assert (o['verts'].cpu() - ref['verts']).abs().max() <= 1e-7
    assert (o['msdf'].cpu() - ref['msdf']).abs().max() <= 1e-7
    # hmSDF "body" variant at full size
    ob = mtets.marching_tets(v.cuda(), sdf.cuda(), msdf.cuda(), t.cuda(), body=True)
    refb = OMT.gshell_tets(v, sdf, msdf, t, negate_msdf=True)
    assert torch.equal(ob['faces'].cpu(), refb['faces'])


def test_full_grid_sdf_sweep_vs_oracle_sample_and_linearity(gpu, grid63):
    from conftest import golden
    from d3h import sdf_mlp
    from oracle import sdf_mlp as O
    g = golden('sdf_mlp.npz')
    keys = sdf_mlp._PARAM_ORDER
    params = [torch.from_numpy(g['sd.net.' + k]).cuda() for k in keys]
    v, _ = grid63
    deform = (torch.rand(v.shape, generator=torch.Generator().manual_seed(2)) * 2 - 1)
    disp = 1.0 / 126 / 2.1
    sdf = sdf_mlp.sdf_query(v.cuda(), params, deform=deform.cuda(), disp=disp).reshape(-1).cpu()
    assert sdf.shape[0] == 262144 and torch.isfinite(sdf).all()
    idx = torch.randperm(v.shape[0], generator=torch.Generator().manual_seed(3))[:4096]
    sd = {('net.' + k): torch.from_numpy(g['sd.net.' + k]) for k in keys}
    ref = O.mlp_forward((v + disp * deform)[idx], sd).reshape(-1)
    assert (sdf[idx] - ref).abs().max() < 2e-7
    m = ref.abs() > 1e-6
    assert torch.equal(sdf[idx][m] > 0, ref[m] > 0)
    # backward is linear in the upstream gradient: grads(2 g1 + g2) == 2 grads(g1) + grads(g2) on a 64k-point slice
    x = v[:65536].cuda()
    ps = [p.clone().requires_grad_(True) for p in params]
    gen = torch.Generator().manual_seed(4)
    g1, g2 = torch.randn(65536, 1, generator=gen).cuda(), torch.randn(65536, 1, generator=gen).cuda()

    def grads(go):
        for p in ps:
            p.grad = None
        (sdf_mlp.sdf_query(x, ps) * go).sum().backward()
        return [p.grad.clone() for p in ps]
    a, b, c = grads(g1), grads(g2), grads(2 * g1 + g2)
    for ga, gb, gc in zip(a, b, c):
        assert (gc - (2 * ga + gb)).abs().max() <= 2e-3 * gc.abs().max() + 1e-6


def test_x3_sweeps_match_exact_f32_mfma_full_size(gpu, grid63, monkeypatch):
    """the bf16 x 3 sweeps (csrc/sdf_mlp_x3.h: forward with save, first-order backward through the active-tile list, the eikonal chain:
    forward, gradient, tangent, injected reverse) against the exact-f32 MFMA kernels on the same 262 144 grid vertices / 50 000 samples:
    values to fp32 rounding, signs equal wherever |sdf| > 1e-7 (what decides the extracted triangles), gradients to 1e-4 of their maximum
    (atomic-order noise of the weight-gradient flush included)"""
    from conftest import golden
    from d3h import sdf_mlp
    g = golden('sdf_mlp.npz')
    keys = sdf_mlp._PARAM_ORDER
    v, _ = grid63
    x = v.cuda()
    deform = ((torch.rand(v.shape, generator=torch.Generator().manual_seed(2)) * 2 - 1)).cuda()
    disp = 1.0 / 126 / 2.1
    go = torch.zeros(x.shape[0], 1, device='cuda')
    sel = torch.randperm(x.shape[0], generator=torch.Generator().manual_seed(5))[:30000].cuda()
    go[sel] = torch.randn(sel.shape[0], 1, generator=torch.Generator().manual_seed(6)).cuda()
    pts = (torch.rand(50000, 3, generator=torch.Generator().manual_seed(7)) * 1.6 - 0.8).cuda()
    res = {}
    for x3 in (True, False):
        monkeypatch.setattr(sdf_mlp, 'X3', x3)
        ps = [torch.from_numpy(g['sd.net.' + k]).cuda().requires_grad_(True) for k in keys]
        dfm = deform.clone().requires_grad_(True)
        pk = sdf_mlp.PackedWeights(ps)
        assert (pk.wpf is not None) == x3
        sdf = sdf_mlp.sdf_query(x, ps, deform=dfm, disp=disp, pack=pk)
        eik = sdf_mlp.eikonal_loss(pts, ps, 0.05, pack=pk)
        ((sdf * go).sum() + eik).backward()
        res[x3] = (sdf.detach().reshape(-1), float(eik), dfm.grad.clone(), [p.grad.clone() for p in ps])
    s3, e3, d3, g3 = res[True]
    s1, e1, d1, g1 = res[False]
    assert (s3 - s1).abs().max() <= 1e-7 + 2e-7 * s1.abs().max(), (s3 - s1).abs().max()
    m = s1.abs() > 1e-7
    assert torch.equal(s3[m] > 0, s1[m] > 0)
    assert abs(e3 - e1) <= 1e-6 * abs(e1), (e3, e1)
    assert (d3 - d1).abs().max() <= 1e-4 * d1.abs().max()
    for k, a, b in zip(keys, g3, g1):
        assert (a - b).abs().max() <= 1e-4 * b.abs().max() + 1e-9, (k, float((a - b).abs().max()), float(b.abs().max()))


def test_full_resolution_raster_properties(gpu):
    """1024^2 x 4: ids in range, interpolation of a constant attribute is the coverage mask, antialias preserves constant images and
    only changes pixels next to an id discontinuity, rasterize is invariant to a permutation of the batch"""
    from d3h import raster, mtets, synth
    v, t = (torch.from_numpy(a) for a in synth.kuhn_grid(24))
    sdf = synth.body_sdf(v)
    o = mtets.marching_tets(v.cuda(), sdf.cuda(), torch.ones(v.shape[0]).cuda(), t.cuda())
    verts, tri = o['verts'], o['faces32']
    mv, mvp, campos = synth.camera(1024)
    B = 4
    offs = torch.tensor([[0.02 * b, 0.0, 0.0] for b in range(B)]).cuda()
    vh = torch.cat([verts[None] + offs[:, None], torch.ones(B, verts.shape[0], 1).cuda()], -1)
    clip = vh @ torch.from_numpy(mvp).cuda().T
    rast, db = raster.rasterize(clip.contiguous(), tri, (1024, 1024))
    ids = rast[..., 3]
    assert ids.min() >= 0 and ids.max() <= tri.shape[0] and (ids > 0).float().mean() > 0.03
    assert (rast[..., 0] >= -1e-4).all() and (rast[..., 1] >= -1e-4).all() and (rast[..., 0] + rast[..., 1] <= 1 + 1e-4).all()
    perm = [2, 0, 3, 1]
    rast_p, _ = raster.rasterize(clip[perm].contiguous(), tri, (1024, 1024))
    assert torch.equal(rast_p, rast[perm])
    ones = torch.ones(1, verts.shape[0], 2).cuda()
    out, _ = raster.interpolate(ones, rast, tri)
    assert torch.allclose(out[..., 0], (ids > 0).float(), atol=1e-5)
    const = torch.full((B, 1024, 1024, 3), 0.37).cuda()
    assert torch.equal(raster.antialias(const, rast, clip.contiguous(), tri), const)
    col = (ids > 0).float()[..., None].expand(-1, -1, -1, 3).contiguous()
    aa = raster.antialias(col, rast, clip.contiguous(), tri)
    changed = (aa != col).any(-1)
    edge = torch.zeros_like(changed)
    edge[:, :, 1:] |= ids[:, :, 1:] != ids[:, :, :-1]
    edge[:, :, :-1] |= ids[:, :, 1:] != ids[:, :, :-1]
    edge[:, 1:, :] |= ids[:, 1:, :] != ids[:, :-1, :]
    edge[:, :-1, :] |= ids[:, 1:, :] != ids[:, :-1, :]
    assert changed.any() and not (changed & ~edge).any()
    assert (aa >= -1e-6).all() and (aa <= 1 + 1e-6).all()


def test_seq_stage_step_and_offset_network_gradient_on_gpu(gpu):
    """seq stage (reduced size): a few iterations with the reference's term weights (train.py:1412-1421) stay finite and move the
    offsets; on a fixed batch the end-to-end gradient w.r.t. the non-rigid network equals the one autograd gives through the library
    formulation of that network."""
    from d3h.scene import Scene
    sc = Scene(res=512, grid_n=32, n_frames=1, device='cuda', prefit_steps=0, loss_set='seq', body_verts=4096)
    for i in range(4):
        r = sc.step_seq()
        assert all(torch.isfinite(v).all() for v in r.values())
    assert float(r['delta_loss']) > 0.0
    for k in ('laplacian_loss', 'nds_normal_loss', 'colli_loss'):
        assert float(r[k]) >= 0.0
    g = sc.geometry
    bg = torch.rand(1, 512, 512, 3, device='cuda')

    def terms():
        torch.manual_seed(0)
        tgt = sc.target(bg)
        tgt.update({'cloth_img': sc.cloth_img, 'body_img': sc.body_img})
        return g.tick_seq(sc.glctx, tgt, None, sc.material, sc.loss_fn, 5, None, t='all')

    def regularisers():
        t = terms()
        return 1000000 * t['laplacian_loss'] + 100000 * t['colli_loss'] + 1000 * t['nds_normal_loss'] + t['delta_loss']

    def smooth_terms():
        t = terms()
        return 1000 * t['nds_normal_loss'] + t['delta_loss']

    def total():
        t = terms()
        return 250 * t['normal_loss'] + 0.1 * t['reg_loss'] + (t['body_msk_loss'] + t['cloth_msk_loss'] + t['all_msk_loss']) + regularisers()
    params = list(g.nonrigid.parameters()) + [g.fix_code]
    # The end-to-end gradient w.r.t. the offset network, checked against autograd through the library formulation of the same network
    # (MLP_deform.forward_reference) at the same point -- everything downstream (LBS, mesh terms) identical.  A first-order descent
    # check like the init-stage test's is not reliable here: through the beta = 100 softplus network the measured decrease of these
    # terms scattered between 0.14x and 2x the prediction at every step length tried, the stiff 1e6-weighted Laplacian sum leaves a
    # usable step whose predicted decrease sits at the fp32 resolution of the total, and the collision term is piecewise.
    def grads(fused):
        g.nonrigid.fused = fused
        for p in params:
            p.grad = None
        smooth_terms().backward()
        return torch.cat([p.grad.reshape(-1) for p in params]).double()
    was = g.nonrigid.fused
    assert was, 'the seq scene must run the fused offset network'
    try:
        ga, gb = grads(True), grads(False)
    finally:
        g.nonrigid.fused = was
    assert torch.isfinite(ga).all() and float(gb.norm()) > 0
    cos = float(ga @ gb / (ga.norm() * gb.norm()))
    assert cos > 0.9995 and 0.99 < float(ga.norm() / gb.norm()) < 1.01, (cos, float(ga.norm()), float(gb.norm()))
    assert torch.isfinite(regularisers())
    # (at this step length the rasterised terms of `total` change by less than their run-to-run noise -- unordered atomics in the
    # image-space backward, discrete coverage -- so only its value is checked here; their gradients are covered by the init-stage test)
    assert torch.isfinite(total())


def test_full_resolution_image_space_passes_vs_torch(gpu):
    """composite, fused per-pixel losses and tiled SSIM at the benchmark's image size (4 x 1024^2) against the same formulas written
    with torch ops on the GPU (fp32): values and the gradient w.r.t. the inputs"""
    import torch.nn.functional as F
    import parity_cases as PC
    from d3h import imgops as I
    from oracle import image_ops as O
    dev = 'cuda'
    B, H, W = 4, 1024, 1024
    g = torch.Generator(device=dev).manual_seed(3)
    rast = torch.zeros(B, H, W, 4, device=dev)
    yy, xx = torch.meshgrid(torch.arange(H, device=dev), torch.arange(W, device=dev), indexing='ij')
    inside = ((xx - 512) ** 2 / 300.0 ** 2 + (yy - 540) ** 2 / 420.0 ** 2) < 1
    rast[..., 3] = inside.float() * 7
    kd = torch.rand(B, H, W, 6, device=dev, generator=g).requires_grad_(True)
    gn = torch.nn.functional.normalize(torch.randn(B, H, W, 3, device=dev, generator=g), dim=-1).requires_grad_(True)
    packed = (torch.rand(B, H, W, 10, device=dev, generator=g) * 2 - 1).requires_grad_(True)
    bg = torch.rand(B, H, W, 3, device=dev, generator=g)
    cref = torch.rand(B, H, W, 4, device=dev, generator=g)
    cref[..., 3] = (((xx - 520) ** 2 / 310.0 ** 2 + (yy - 530) ** 2 / 415.0 ** 2) < 1).float()
    nref = torch.nn.functional.normalize(torch.randn(B, H, W, 3, device=dev, generator=g), dim=-1) * cref[..., 3:]
    cov = (rast[..., 3:] > 0).float()

    def fused():
        st = I.composite(rast, [(kd[..., 0:3], I.COMP_IMAGE, bg), (gn, I.COMP_ZERO, None), (packed[..., 9:10], I.COMP_ALPHA, None)])
        pl = I.pixel_losses(st, {'shaded': (0, 4), 'geometric_normal': (4, 4), 'msdf_image': (8, 1)}, cref, nref, ('l1', 'log_srgb'), want_ssim=True)
        return torch.stack([pl[k] for k in ('mask_mse', 'img', 'msdf_pos_l1', 'msdf_neg_l1', 'normal_mse', 'normal_cos', 'ssim')])

    def plain():
        sh = torch.where(cov > 0, torch.cat((kd[..., 0:3], torch.ones_like(cov)), -1), torch.cat((bg, torch.zeros_like(cov)), -1))
        gno = torch.cat((gn, torch.ones_like(cov)), -1) * cov
        mi = cov * packed[..., 9:10]
        gm = cref[..., 3:]
        out_n = F.normalize(gno[..., 0:3], p=2, dim=-1) * torch.tensor([1.0, -1.0, -1.0], device=dev)
        gt_n = F.normalize(nref, p=2, dim=-1)
        a, b = (sh[..., 0:3] * gm).permute(0, 3, 1, 2), (cref[..., 0:3] * gm).permute(0, 3, 1, 2)
        return torch.stack([F.mse_loss(sh[..., 3:], gm), O.image_loss(sh[..., 0:3] * gm, cref[..., 0:3] * gm, 'l1', 'log_srgb'),
                            F.l1_loss(mi.clamp(min=0) * (gm == 0).float(), torch.zeros_like(gm)),
                            F.l1_loss(mi.clamp(max=0) * (gm == 1).float(), torch.ones_like(gm)), F.mse_loss(out_n, gt_n),
                            F.cosine_similarity(out_n.reshape(-1, 3), gt_n.reshape(-1, 3), dim=1).mean(), O.ssim(a, b)])
    w = torch.tensor([100.0, 1.0, 0.5, 0.5, 1.0, -0.1, -1.0], device=dev)
    vf = fused()
    (vf * w).sum().backward()
    gf = [t.grad.clone() for t in (kd, gn, packed)]
    for t in (kd, gn, packed):
        t.grad = None
    vp = plain()
    (vp * w).sum().backward()
    for name, a_, b_ in zip(('mask', 'img', 'msdf+', 'msdf-', 'nmse', 'ncos', 'ssim'), vf.tolist(), vp.tolist()):
        assert abs(a_ - b_) <= 1e-6 + 3e-5 * abs(b_), (name, a_, b_)
    for name, a_, t in zip(('kd', 'gn', 'packed'), gf, (kd, gn, packed)):
        den = float(t.grad.abs().max())
        assert den > 0 and float((a_ - t.grad).abs().max()) <= 2e-4 * den, (name, float((a_ - t.grad).abs().max()), den)


def test_split_stage_shared_sweep_matches_two_sweeps(gpu):
    """Scene.step_split evaluates the SDF sweep once for the garment and the body extraction of an iteration (FLAGS.share_sdf_sweep);
    the reference runs it twice (hmsdf.py:527-538 per tick_split).  Same losses and the same parameter gradients either way."""
    from d3h.scene import Scene
    res = {}
    sc = Scene(res=256, grid_n=32, n_frames=2, device='cuda', prefit_steps=300, loss_set='split')
    for share in (True, False):
        # the LambdaLR warm-up makes the first update a no-op (lr = 0) and the second iteration's gradients are taken before its
        # update: both passes see the same parameters; same seed -> same backgrounds, jitter and eikonal samples
        sc.share_sweep = share
        sc.it = 0                                         # same schedule weights (sdf_regularizer ramp) in both passes
        torch.manual_seed(77)
        last = sc.step_split()
        g = sc.geometry
        res[share] = ({k: float(v) for k, v in last.items()},
                      [p.grad.detach().clone() for p in g.sdf_net.parameters()] + [g.deform.grad.detach().clone(), g.msdf.grad.detach().clone()])
    la, lb = res[True][0], res[False][0]
    for k in la:
        assert abs(la[k] - lb[k]) <= 2e-4 * max(1e-3, abs(lb[k])), (k, la[k], lb[k])
    for a, b in zip(res[True][1], res[False][1]):
        assert (a - b).abs().max() <= 2e-3 * b.abs().max() + 1e-9


def _oracle_mesh_check(g, d, body=False, min_faces=5000):
    """the mesh a tick extracted == oracle marching tets on that tick's own inputs (deformed grid, sdf, msdf), bit-exact indices"""
    from oracle import marching_tets as OMT
    with torch.no_grad():
        v_def = (g.verts + g.max_displacement * g.deform).detach().cpu()
        ref = OMT.gshell_tets(v_def, d['sdf'].detach().cpu().reshape(-1), g.msdf.detach().cpu(), g.indices.cpu(), negate_msdf=body)
    faces = d['imesh'].t_pos_idx.cpu().long()
    assert faces.shape[0] >= min_faces
    assert torch.equal(faces, ref['faces']), 'faces differ from the oracle'
    assert (d['imesh'].v_pos.detach().cpu() - ref['verts']).abs().max() <= 1e-6
    return faces.shape[0]


def test_config3_full_size_tick_and_steps(gpu):
    """BASELINE config 3 exactly as bench.py runs it (4 frames, tet-res 128 = Kuhn n 63, 1024^2, mask + normal + SSIM + sdf_reg + eikonal,
    FLAGS.visualize_watertight on): a forward tick whose extracted mesh is bit-exact against the oracle on the same sdf, then
    optimiser steps with finite losses / gradients and a falling total"""
    from d3h.scene import Scene
    sc = Scene(res=1024, grid_n=63, n_frames=4, device='cuda', prefit_steps=300, loss_set='full', visualize_watertight=True)
    g = sc.geometry
    assert g.verts.shape[0] == 262144 and g.indices.shape[0] == 1500282
    bg = torch.rand(4, 1024, 1024, 3, device='cuda')
    r = g.tick_init(sc.glctx, sc.target(bg), None, sc.material, sc.loss_fn, 0, None)
    assert all(torch.isfinite(v).all() for v in r.values())
    d = g.last_mesh_dict
    nf = _oracle_mesh_check(g, d)
    b = d['buffers']
    assert b['shaded'].shape == (4, 1024, 1024, 4) and b['geometric_normal'].shape == (4, 1024, 1024, 4) and b['msdf_image'].shape == (4, 1024, 1024, 1)
    assert 0.03 < float((b['shaded'][..., 3] > 0.5).float().mean()) < 0.6
    # inside a tick the watertight twin is not rendered (no loss reads it and ticks return loss values only); render_init called directly
    # -- what validate_itr does -- honours FLAGS.visualize_watertight, as does the reference-equivalent 'all' mode of the tick
    assert 'buffers_watertight' not in d
    with torch.no_grad():
        dv = g.render_init(sc.glctx, sc.target(bg), None, sc.material)
    assert 'buffers_watertight' in dv and dv['buffers_watertight']['shaded'].shape == (4, 1024, 1024, 4) and 'depth' in dv['buffers_watertight']
    sc.FLAGS.render_buffers = 'all'
    r_all = g.tick_init(sc.glctx, sc.target(bg), None, sc.material, sc.loss_fn, 0, None)
    assert 'buffers_watertight' in g.last_mesh_dict and len(g.last_mesh_dict['buffers']['_layout']) == 3
    for k in ('msk_loss', 'img_loss', 'normal_loss'):
        assert abs(float(r_all[k]) - float(r[k])) <= 1e-5 * max(1e-6, abs(float(r[k]))), k
    sc.FLAGS.render_buffers = None
    hist = []
    for i in range(12):
        out = sc.step()
        assert all(torch.isfinite(v).all() for v in out.values()), (i, out)
        for p in list(g.parameters()) + list(sc.material['kd_ks'].parameters()):
            assert p.grad is None or torch.isfinite(p.grad).all()
        hist.append(float(out['total']))
    gmax = max(float(p.grad.abs().max()) for p in g.sdf_net.parameters())
    print(f'config 3 full size: {nf} faces; total {hist[0]:.4f} -> {hist[-1]:.4f}; max |d total / d sdf weights| {gmax:.2e}')
    assert gmax < 1e4
    assert hist[-1] < hist[0]


def test_config5_shape_split_stage_full_size(gpu):
    """BASELINE config 5's per-GPU work (dual body + garment extraction with hmSDF_Tets, tet-res 128, 1024^2, the split stage's loss
    stack; one frame on this GPU): both extractions bit-exact against the oracle, steps finite"""
    from d3h.scene import Scene
    import lpips
    from conftest import golden
    lp = lpips.LPIPS(net='alex', pretrained=False)                        # config 5: "full loss stack incl. LPIPS"
    gl = golden('lpips.npz')
    lp.load_state_dict({f'lin{k}.model.1.weight': torch.from_numpy(gl[f'alex.lin{k}']) for k in range(5)}, strict=False)
    sc = Scene(res=1024, grid_n=63, n_frames=1, device='cuda', prefit_steps=300, loss_set='split', lpips=lp)
    g = sc.geometry
    bg = torch.rand(1, 1024, 1024, 3, device='cuda')
    for typ in ('cloth', 'body'):
        r = g.tick_split(sc.glctx, sc.target(bg), None, sc.material, sc.loss_fn, 0, None, type=typ)
        assert all(torch.isfinite(v).all() for v in r.values() if torch.is_tensor(v)), (typ, r)
        # (Scene's split-stage partition: the mSDF is positive on the torso band -- the garment -- and negative elsewhere, so BOTH passes
        # extract a real mesh; with the reference's initial mSDF, hmsdf.py:311, the body pass would see a sliver)
        _oracle_mesh_check(g, g.last_mesh_dict, body=(typ == 'body'), min_faces=3000)
    assert min(sc.split_faces.values()) >= 3000, sc.split_faces
    hist = []
    for i in range(6):
        out = sc.step_split()
        assert all(torch.isfinite(v).all() for v in out.values()), (i, out)
        hist.append(float(out['total']))
    assert torch.isfinite(sc.material['kd_ks'].encoder.params.grad).all()
    assert float(g.last_lpips_loss) > 0                                  # the LPIPS term is live in the split stage's image loss
    print(f'config 5 shape (1 GPU): total {hist[0]:.4f} -> {hist[-1]:.4f}; LPIPS term {float(g.last_lpips_loss):.4f}')


# ---- one WHOLE tick at BASELINE sizes against the oracle chain on the same state (oracle/parity.py; the oracle uses its own rasteriser) -------
def _fail(what, key, value, bar):
    """one short line: the failing quantity, its value, its bar (the round-5 assertion message was 16 KB)"""
    return '%s %s = %.3g > bar %.3g' % (what, key, value, bar)


def _check_tick_parity(rep, n_grid):
    """Bars of the whole-tick comparison (profiles/r6_parity_rootcause.md has the measurements behind every number).  The state is
    reproducible (tests/golden/parity_state_sdf.npz + seeded host-side fields), so these figures are the same on every box up to the
    ~1e-6 noise of the float atomics in the GPU backward.

    Exact: triangle indices (bit-exact).  Discrete decisions of the two rasterisers are counted, not tolerated away: pixels won by another
    triangle (`raster_ids_differ`: a pixel centre within rounding of an interior edge), antialias pairs decided differently
    (`alpha_pixels_differ`), texture gates within rounding of zero (`relu_kinks`); the grid vertices behind the triangles they sit on are excluded
    from the per-vertex comparison explicitly and counted (oracle/parity.py:kink_grid_vertices) -- and there must be none of them when no
    decision was counted.

    With the product's per-pixel winners shared (everything downstream of the discrete pass):
      * every loss term 1e-3;
      * per-grid-vertex tensors (deform, msdf): relative L2 2e-3 (mask-only) / 5e-3 (full loss set), max-norm 5e-2 of the tensor's largest
        entry, at most 8 vertices above 2e-3 of it;
      * tensors that sum over all pixels (SDF weights and biases, trans, texture), no counted decision: 2e-3 in the full loss set (measured over
        four seeds: weights <= 1.5e-3; the biases -- plain cancelling sums of dL/dsdf -- 4e-3, measured <= 2.3e-3, where the fp32 oracle itself
        is up to 1.7e-3 from the float64 evaluation); 5e-3 (biases 1e-2) in the mask-only tick, whose whole SDF gradient passes through the antialias position gradient of a few hundred silhouette pixel
        pairs -- the worst-conditioned path of the tick in fp32: the float64 evaluation of the oracle chain puts the reference's own fp32
        arithmetic 7e-4 / 1.3e-3 away from the exact gradient there.  With counted decisions: 2e-2 / 5e-2.
    Against the float64 evaluation of the oracle chain (`float64`, when present): the GPU tick may be at most 5 x as far from it as the fp32
    oracle tick is, with a floor of 1e-3 (measured 0.6 ... 4.0 x).
    With NOTHING shared: per-vertex tensors 5e-3 in L2 after excluding the triangles of the pixels the two rasterisers give to different
    winners; all-pixel sums 5e-2 (they contain those triangles: measured <= 8.7e-3 with 70 such pixels)."""
    assert rep['mesh_faces_equal'], 'extracted triangle indices differ from the oracle at full size'
    assert rep['raster_ids_differ'] <= max(3, rep['pixels'] // 5000), _fail('own', 'raster_ids_differ', rep['raster_ids_differ'], max(3, rep['pixels'] // 5000))
    assert rep['alpha_pixels_differ'] <= max(5, rep['pixels'] // 50000), _fail('own', 'alpha_pixels_differ', rep['alpha_pixels_differ'], max(5, rep['pixels'] // 50000))
    sh, own = rep['shared_raster'], rep['own_raster']
    assert sh['alpha_pixels_differ'] <= max(2, rep['pixels'] // 65536), _fail('shared', 'alpha_pixels_differ', sh['alpha_pixels_differ'], max(2, rep['pixels'] // 65536))
    assert sh['max_rel_loss_diff'] <= 1e-3, _fail('shared', 'loss', sh['max_rel_loss_diff'], 1e-3)
    kinks = rep['relu_kinks'] + sh['alpha_pixels_differ']
    if kinks == 0:
        assert sh['excluded_grid_vertices'] == 0, 'grid vertices were excluded although no discrete decision was counted'
    assert sh['excluded_grid_vertices'] <= max(400, n_grid // 40), _fail('shared', 'excluded_grid_vertices', sh['excluded_grid_vertices'], max(400, n_grid // 40))
    mask_only = 'loss set "mask"' in rep['config']
    for which in ('max_rel_grad_diff_excl', 'l2_rel_grad_diff_excl'):
        is_max = which.startswith('max')
        for k, v in sh[which].items():
            if v is None:
                continue
            if k in ('deform', 'msdf'):
                bar = 5e-2 if is_max else (2e-3 if mask_only else 5e-3)
            elif kinks > 0:
                bar = 5e-2 if k == 'sdf_net_bias' else 2e-2
            elif mask_only:
                bar = 1e-2 if k == 'sdf_net_bias' else 5e-3
            else:
                bar = 4e-3 if k == 'sdf_net_bias' else 2e-3
            assert v <= bar, _fail('shared ' + which, k, v, bar)
    assert all(v <= 8 for v in sh['vertex_outliers_excl'].values()), 'vertex outliers %s' % sh['vertex_outliers_excl']
    f64 = rep.get('float64')
    if f64 is not None:
        assert f64['same_mesh'], 'the float64 oracle run extracted another mesh (a sign decision within float32 rounding of zero): pick another seed'
        assert f64['gpu_loss'] <= 1e-3, _fail('vs float64', 'gpu loss', f64['gpu_loss'], 1e-3)
        for side in ('l2', 'max'):
            for k, v in f64['gpu_' + side].items():
                o = f64['oracle32_' + side][k]
                if v is None or o is None:
                    continue
                bar = max(5.0 * o, 1e-3)
                assert v <= bar, _fail('gpu vs float64 (%s; the fp32 oracle: %.3g)' % (side, o), k, v, bar)
    assert own['max_rel_loss_diff'] <= 2e-3, _fail('own', 'loss', own['max_rel_loss_diff'], 2e-3)
    assert own['excluded_grid_vertices'] <= max(2000, n_grid // 20), _fail('own', 'excluded_grid_vertices', own['excluded_grid_vertices'], max(2000, n_grid // 20))
    for which in ('max_rel_grad_diff_excl', 'l2_rel_grad_diff_excl'):
        for k, v in own[which].items():
            if v is None:
                continue
            bar = 5e-2 if k not in ('deform', 'msdf') else (5e-3 if which.startswith('l2') or mask_only else 5e-2)
            assert v <= bar, _fail('own ' + which, k, v, bar)


PARITY_STATE = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'parity_state_sdf.npz')


def _fixed_scene(res, grid_n, loss_set, seed):
    """a REPRODUCIBLE scene: the SDF network is the CPU-fitted fixture tests/golden/parity_state_sdf.npz
    (tools/gen_parity_state.py), deform / trans / mSDF / texture are seeded host-side fields (Scene.perturb_state_seeded,
    set_kinkfree_texture) -- no pre-fit and no optimiser step on the GPU (their float atomics made every box compare a different scene)"""
    from d3h import scene
    sc = scene.Scene(device='cuda', visualize_watertight=True, res=res, grid_n=grid_n, n_frames=1, loss_set=loss_set, sdf_state=PARITY_STATE)
    sc.perturb_state_seeded(seed)
    if loss_set != 'mask':
        sc.set_kinkfree_texture(seed)
    return sc


def _parity_line(rep):
    sh, own = rep['shared_raster'], rep['own_raster']
    f = lambda d: {k: (None if v is None else float('%.2g' % v)) for k, v in d.items()}
    return ('faces %d eq %s ids_differ %d alpha own/shared %d/%d relu %d | shared loss %.1e max %s l2 %s excl %d | own loss %.1e l2 %s excl %d' % (
        rep['mesh_faces'], rep['mesh_faces_equal'], rep['raster_ids_differ'], rep['alpha_pixels_differ'], sh['alpha_pixels_differ'], rep['relu_kinks'],
        sh['max_rel_loss_diff'], f(sh['max_rel_grad_diff_excl']), f(sh['l2_rel_grad_diff_excl']), sh['excluded_grid_vertices'],
        own['max_rel_loss_diff'], f(own['l2_rel_grad_diff_excl']), own['excluded_grid_vertices'])) + \
        ('' if 'float64' not in rep else ' | vs float64: gpu l2 %s max %s ; oracle32 l2 %s max %s ; same mesh %s' % (
            f(rep['float64']['gpu_l2']), f(rep['float64']['gpu_max']), f(rep['float64']['oracle32_l2']), f(rep['float64']['oracle32_max']), rep['float64']['same_mesh']))


@pytest.mark.timeout(600)
@pytest.mark.slow
def test_whole_tick_config2_full_size_vs_oracle(gpu):
    """BASELINE configs[1] in full -- 1 frame, tet-res 64 (35 937 vertices / 196 608 tets), 512 x 512, 50 000 eikonal samples: every loss
    term and d(msk + reg)/d{16 SDF tensors, deform, msdf, trans, grid table, texture MLP} of one GPU tick against the oracle tick on the
    same, reproducible state (the comparison bench.py reports as cpu_baseline.parity)."""
    from oracle import parity as OP
    sc = _fixed_scene(512, 32, 'mask', seed=0)
    rep, _ = OP.scene_tick_parity(sc, iteration=10, seed=0, truth64=True)
    print('config2 whole tick:', _parity_line(rep))
    assert rep['mesh_faces'] > 1000 and rep['relu_kinks'] == 0
    _check_tick_parity(rep, sc.geometry.verts.shape[0])


@pytest.mark.timeout(1500)
@pytest.mark.slow
def test_whole_tick_config3_shape_one_frame_vs_oracle(gpu):
    """the config-3 loss stack (mask + normal + SSIM + sdf_reg + eikonal) at tet-res 128 (262 144 vertices / 1 500 282 tets), 1024 x 1024,
    ONE frame (the oracle's CPU rasteriser and its autograd graph bound the size): one GPU tick against the oracle tick on the same,
    reproducible state"""
    from oracle import parity as OP
    sc = _fixed_scene(1024, 63, 'full', seed=1)
    rep, tm = OP.scene_tick_parity(sc, iteration=10, seed=1, truth64=True)
    print('config3-shape whole tick:', _parity_line(rep), '| oracle fwd %.0f s bwd %.0f s' % (tm['forward_s'], tm['backward_s']))
    assert rep['mesh_faces'] > 5000 and rep['relu_kinks'] == 0
    _check_tick_parity(rep, sc.geometry.verts.shape[0])


# ---- the reference's own working point: configs/f3c.json -- batch 1, train_res 1080 x 1080 (a multiple of NO tile size the kernels use:
# 64-wide SSIM strips, 16 x 16 / 8 x 8 raster tiles, 256-thread rows), tet grid 128 --------------------------------------------------------
def test_reference_working_point_1080_per_op(gpu):
    """the image-space ops against their oracle / torch formulations at 1 x 1080 x 1080: fused pixel losses + SSIM, composite; rasterizer
    properties (ids in range, barycentrics, constant-image antialias invariance, antialias only next to id discontinuities)"""
    import parity_cases as P
    from d3h import raster, mtets, synth
    P.check_pixel_losses('cuda', B=1, H=1080, W=1080, with_ssim=True)
    P.check_composite('cuda', B=1, H=1080, W=1080)
    v, t = (torch.from_numpy(a) for a in synth.kuhn_grid(24))
    o = mtets.marching_tets(v.cuda(), synth.body_sdf(v).cuda(), torch.ones(v.shape[0]).cuda(), t.cuda())
    verts, tri = o['verts'], o['faces32']
    mv, mvp, campos = synth.camera(1080)
    clip = (torch.cat([verts[None], torch.ones(1, verts.shape[0], 1).cuda()], -1) @ torch.from_numpy(mvp).cuda().T).contiguous()
    rast, db = raster.rasterize(clip, tri, (1080, 1080))
    ids = rast[..., 3]
    assert rast.shape == (1, 1080, 1080, 4) and ids.min() >= 0 and ids.max() <= tri.shape[0] and (ids > 0).float().mean() > 0.03
    assert (rast[..., 0] >= -1e-4).all() and (rast[..., 1] >= -1e-4).all() and (rast[..., 0] + rast[..., 1] <= 1 + 1e-4).all()
    assert (ids[:, -1, :] >= 0).all() and (ids[:, :, -1] >= 0).all()                     # last row / column written (1080 = 16 * 67 + 8)
    const = torch.full((1, 1080, 1080, 3), 0.37).cuda()
    assert torch.equal(raster.antialias(const, rast, clip, tri), const)
    col = (ids > 0).float()[..., None].expand(-1, -1, -1, 3).contiguous()
    aa = raster.antialias(col, rast, clip, tri)
    changed = (aa != col).any(-1)
    edge = torch.zeros_like(changed)
    edge[:, :, 1:] |= ids[:, :, 1:] != ids[:, :, :-1]
    edge[:, :, :-1] |= ids[:, :, 1:] != ids[:, :, :-1]
    edge[:, 1:, :] |= ids[:, 1:, :] != ids[:, :-1, :]
    edge[:, :-1, :] |= ids[:, 1:, :] != ids[:, :-1, :]
    assert changed.any() and not (changed & ~edge).any()
    ones = torch.ones(1, verts.shape[0], 2).cuda()
    out, _ = raster.interpolate(ones, rast, tri)
    assert torch.allclose(out[..., 0], (ids > 0).float(), atol=1e-5)


def test_reference_working_point_1080_steps(gpu):
    """bench.py --config f3c in small: tet-res 128, ONE 1080 x 1080 frame, the init-stage total of train.py:718 with the MobileNetV2-feature
    normal loss (seeded random trunk): a dozen optimiser steps stay finite, every loss term is produced, the mask term goes down"""
    from d3h import scene
    sc = scene.Scene(device='cuda', prefit_steps=200, visualize_watertight=True, res=1080, grid_n=63, n_frames=1, loss_set='init',
                     flags_hook=lambda F: setattr(F, 'use_perceptual_normal_loss', True))
    assert sc.FLAGS.normal_loss_fn is not None and sc.FLAGS.ssim_weight == 0.0
    first = None
    for i in range(12):
        r = sc.step()
        assert all(torch.isfinite(v).all() for v in r.values()), (i, r)
        first = first if first is not None else {k: float(v) for k, v in r.items()}
    assert float(r['normal_loss']) > 0 and float(r['eik_loss']) > 0 and float(r['sdf_reg_loss']) > 0
    assert float(r['msk_loss']) < first['msk_loss']
    b = sc.geometry.last_mesh_dict['buffers']
    assert b['shaded'].shape == (1, 1080, 1080, 4)


@pytest.mark.timeout(600)
def test_marching_tets_beyond_two_million_tets(gpu):
    """a grid LARGER than anything BASELINE names -- Kuhn n = 80: 531 441 vertices, 3 072 000 tets, 3.6 M edges.  Rounds 1-3 returned
    D3H_ERR_ARG above 2 097 152 tets or edges (the single-segment scan of the per-block counters); the scan now walks segments with a
    carry.  Faces bit-exact against the oracle, cloth and body variants"""
    from d3h import mtets, synth
    from oracle import marching_tets as OMT
    v, t = (torch.from_numpy(a) for a in synth.kuhn_grid(80))
    assert t.shape[0] == 3072000
    sdf = synth.body_sdf(v) + 0.002 * torch.sin(37 * v[:, 0]) * torch.cos(29 * v[:, 2])
    g = torch.Generator().manual_seed(2)
    msdf = torch.rand(v.shape[0], generator=g) * 2 - 0.5
    o = mtets.marching_tets(v.cuda(), sdf.cuda(), msdf.cuda(), t.cuda())
    ref = OMT.gshell_tets(v, sdf, msdf, t)
    assert o['faces'].shape[0] > 10000
    assert torch.equal(o['faces'].cpu(), ref['faces']) and torch.equal(o['faces_wt'].cpu(), ref['faces_watertight'])
    assert (o['verts'].cpu() - ref['verts']).abs().max() <= 1e-7
    ob = mtets.marching_tets(v.cuda(), sdf.cuda(), msdf.cuda(), t.cuda(), body=True)
    refb = OMT.gshell_tets(v, sdf, msdf, t, negate_msdf=True)
    assert torch.equal(ob['faces'].cpu(), refb['faces'])
