"""Generate tests/golden/*.npz by importing the reference (dev container only) and check the oracle
against it.  Run: python tools/gen_golden.py [names...].  The fixtures are DATA (seeded inputs and the
reference's outputs); no reference source is stored.
"""
import os, sys
import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tools'))
sys.path.insert(0, ROOT)
import refharness
from oracle import marching_tets as OMT
from oracle import sdf_mlp as OMLP


def _load_by_path(name, rel):
    """build-side helper modules that only generate INPUTS (synthetic body model, Kuhn grid, camera; the torchvision-shaped MobileNetV2
    trunk) are loaded by file path: putting d3human-code_amd/ on sys.path would make `render`, `geometry`, `deform` resolve to the
    build instead of the reference"""
    import importlib.util
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, 'd3human-code_amd', rel))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


synth = _load_by_path('_d3h_synth_inputs', 'd3h/synth.py')


def _load_pkg(name, rel):
    return _load_by_path(name, rel)

GOLD = os.path.join(ROOT, 'tests', 'golden')
os.makedirs(GOLD, exist_ok=True)


def npy(d):
    return {k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in d.items()}


def gen_sdf_mlp():
    with refharness.ref_ctx():
        from geometry.mlp import MLP
        torch.manual_seed(0)
        m = MLP(skip_in=[3], n_freq=6, n_hidden=6, d_hidden=256)
        g = torch.Generator().manual_seed(1)
        x = (torch.rand(1024, 3, generator=g) * 2.4 - 1.2).requires_grad_(True)
        y = m(x)
        wgt = torch.linspace(-1, 1, 1024)[:, None]
        (y * wgt).sum().backward()
        sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
        out = {'x': x.detach(), 'sdf': y.detach(), 'gout': wgt, 'dx': x.grad}
        for k, p in m.named_parameters():
            out['grad.' + k] = p.grad
        for k, v in sd.items():
            out['sd.' + k] = v
    # oracle check
    yo = OMLP.mlp_forward(x.detach(), sd)
    assert torch.equal(yo, y.detach()) or (yo - y.detach()).abs().max() < 1e-7, (yo - y).abs().max()
    print('sdf_mlp: oracle max diff', (yo - y.detach()).abs().max().item())
    np.savez_compressed(os.path.join(GOLD, 'sdf_mlp.npz'), **npy(out))


def _mtets_inputs(n, seed, mixed_msdf, shuffle):
    verts, tets = OMT.kuhn_grid(n, shuffle_seed=(seed if shuffle else None))
    g = torch.Generator().manual_seed(seed)
    v = torch.from_numpy(verts)
    t = torch.from_numpy(tets)
    # noisy ellipsoid, positive outside
    c = torch.tensor([0.0, -0.2, 0.0])
    r = torch.tensor([0.55, 0.8, 0.45])
    sdf = (((v - c) / r).norm(dim=-1) - 1.0) * 0.4 + 0.02 * torch.randn(v.shape[0], generator=g)
    if mixed_msdf:
        msdf = torch.rand(v.shape[0], generator=g) * 2 - 0.8
    else:
        msdf = (torch.rand(v.shape[0], generator=g) - 0.01).clamp(-1, 1)      # hmsdf.py:311
    pos = v + 0.01 * torch.randn(v.shape, generator=g)
    return pos, sdf, msdf, t


def _run_ref_mtets(cls_name, pos, sdf, msdf, tets, typ=None):
    with refharness.ref_ctx():
        if cls_name == 'gshell':
            from geometry.gshell_tets import GShell_Tets as C
        else:
            from geometry.hmsdf_tets_split import hmSDF_Tets as C
        mt = C()
        pos = pos.clone().requires_grad_(True); sdf = sdf.clone().requires_grad_(True); msdf = msdf.clone().requires_grad_(True)
        if typ is None:
            verts, faces, _, _, v_tng, extra = mt(pos, sdf[:, None], msdf, tets)
        else:
            verts, faces, _, _, v_tng, extra = mt(pos, sdf[:, None], msdf, tets, typ)
        out = {'verts': verts, 'faces': faces, 'v_tng': v_tng, **{k: v for k, v in extra.items()}}
        if verts.shape[0] > 0:
            gen = torch.Generator().manual_seed(7)
            gv = torch.randn(verts.shape, generator=gen)
            gm = torch.randn(extra['msdf'].shape, generator=gen)
            gw = torch.randn(extra['vertices_watertight'].shape, generator=gen)
            loss = (verts * gv).sum() + (extra['msdf'] * gm).sum() + (extra['vertices_watertight'] * gw).sum()
            loss.backward()
            out.update({'g_verts': gv, 'g_msdf': gm, 'g_wt': gw, 'd_pos': pos.grad, 'd_sdf': sdf.grad, 'd_msdf': msdf.grad})
    return out


def gen_mtets():
    cases = [('gshell_n8', 'gshell', 8, 3, False, False, None), ('gshell_n16_mixed', 'gshell', 16, 4, True, True, None),
             ('hmsdf_n8_cloth', 'hmsdf', 8, 5, True, False, 'cloth'), ('hmsdf_n8_body', 'hmsdf', 8, 5, True, False, 'body'),
             ('gshell_n6_empty', 'gshell', 6, 6, False, False, None)]
    for name, cls, n, seed, mixed, shuffle, typ in cases:
        pos, sdf, msdf, tets = _mtets_inputs(n, seed, mixed, shuffle)
        if 'empty' in name:
            sdf = sdf.abs() + 0.1
        ref = _run_ref_mtets(cls, pos, sdf, msdf, tets, typ)
        # oracle check (bit-exact indices, tight fp)
        p2 = pos.clone().requires_grad_(True); s2 = sdf.clone().requires_grad_(True); m2 = msdf.clone().requires_grad_(True)
        o = OMT.gshell_tets(p2, s2, m2, tets, negate_msdf=(typ == 'body'))
        assert torch.equal(o['faces'], ref['faces']), name
        assert torch.equal(o['faces_watertight'], ref['faces_watertight']), name
        assert o['n_verts_watertight'] == ref['n_verts_watertight']
        for k in ('verts', 'v_tng', 'vertices_watertight', 'msdf', 'msdf_watertight', 'msdf_boundary', 'v_tng_watertight'):
            d = (o[k] - ref[k]).abs().max().item() if ref[k].numel() else 0.0
            assert d < 1e-6, (name, k, d)
        if 'd_pos' in ref:
            loss = (o['verts'] * ref['g_verts']).sum() + (o['msdf'] * ref['g_msdf']).sum() + (o['vertices_watertight'] * ref['g_wt']).sum()
            loss.backward()
            for k, t in (('d_pos', p2), ('d_sdf', s2), ('d_msdf', m2)):
                if ref[k] is None:
                    assert t.grad is None, (name, k)
                    continue
                d = (t.grad - ref[k]).abs().max().item()
                rel = d / (ref[k].abs().max().item() + 1e-12)
                assert rel < 1e-4, (name, k, d, rel)
        print(f'mtets {name}: P={ref["verts"].shape[0]} P_wt={ref["n_verts_watertight"]} F={ref["faces"].shape[0]} oracle==reference')
        out = {**{k: v for k, v in ref.items() if v is not None}, 'in_pos': pos, 'in_sdf': sdf, 'in_msdf': msdf, 'tets': tets}
        np.savez_compressed(os.path.join(GOLD, f'mtets_{name}.npz'), **npy(out))


def gen_lbs():
    """reference lbs() + SMPLX_Deformer.{interpolate_weights, apply_lbs_inverse, lbs_forward} on a seeded miniature model.
    The deformer is built with object.__new__ (its __init__ needs the licence-gated SMPL-X files); its `.layer.forward`
    is a small shim that assembles full_pose as body_models.py:1225-1257 does and calls the REFERENCE lbs()."""
    from oracle import lbs as OL
    m = synth.make_body_model(n_verts=512, seed=0, n_shape=10, n_expr=5)
    mt = {k: torch.from_numpy(v) for k, v in m.items()}
    gen = torch.Generator().manual_seed(11)
    nfr = 3
    betas = torch.randn(1, 10, generator=gen) * 0.5
    expr = torch.randn(nfr, 5, generator=gen) * 0.1
    body_pose = torch.randn(nfr, 63, generator=gen) * 0.25
    root_pose = torch.randn(nfr, 3, generator=gen) * 0.2
    jaw = torch.randn(nfr, 3, generator=gen) * 0.05
    hands = torch.randn(nfr, 2, 45, generator=gen) * 0.1      # ignored: full_pose[:, 69:] is zeroed
    eyes = torch.randn(nfr, 2, 3, generator=gen) * 0.05
    trans = torch.randn(nfr, 3, generator=gen) * 0.1
    joint_offset = torch.randn(1, 55, 3, generator=gen) * 0.005
    locator_offset = torch.randn(1, 55, 3, generator=gen) * 0.005
    face_offset = torch.randn(1, 512, 3, generator=gen) * 0.001
    pts = torch.from_numpy(m['v_template'])[torch.randperm(512, generator=gen)[:300]] + 0.03 * torch.randn(300, 3, generator=gen)
    with refharness.ref_ctx():
        from deform.smplx_exavatar.lbs import lbs as ref_lbs
        from deform.smplx_exavatar_deformer import SMPLX_Deformer

        class Layer:
            lbs_weights = mt['weights']

            def forward(self, betas=None, global_orient=None, body_pose=None, jaw_pose=None, leye_pose=None, reye_pose=None,
                        left_hand_pose=None, right_hand_pose=None, expression=None, transl=None, face_offset=None,
                        joint_offset=None, locator_offset=None, pose2rot=True):
                fp = torch.cat([global_orient.reshape(-1, 1, 3), body_pose.reshape(-1, 21, 3), jaw_pose.reshape(-1, 1, 3),
                                leye_pose.reshape(-1, 1, 3), reye_pose.reshape(-1, 1, 3), left_hand_pose.reshape(-1, 15, 3),
                                right_hand_pose.reshape(-1, 15, 3)], dim=1).reshape(-1, 165)
                fp[:, 69:].zero_()
                comp = torch.cat([betas, expression], dim=-1)
                dirs = torch.cat([mt['shapedirs'], mt['expr_dirs']], dim=-1)
                vt = mt['v_template'] if face_offset is None else mt['v_template'] + face_offset
                verts, joints, A = ref_lbs(comp, fp, vt, dirs, mt['posedirs'], mt['J_regressor'], joint_offset, locator_offset,
                                           mt['parents'], mt['weights'], pose2rot=True)
                from types import SimpleNamespace
                return SimpleNamespace(vertices=verts + transl[:, None]), A
            __call__ = forward

        d = object.__new__(SMPLX_Deformer)
        d.layer = Layer(); d.lbs_weights = mt['weights']; d.k = 1; d.expr_param_dim = 5; d.shape_param_dim = 10
        d.layer.faces_tensor = None
        # initialize(): deformer.py:173-238 with the init pose of :178-180
        bp0 = torch.zeros(1, 63); bp0[:, 2] = torch.pi / 36; bp0[:, 5] = -torch.pi / 36
        out0, A0 = d.layer(betas=betas, global_orient=torch.zeros(1, 3), body_pose=bp0, jaw_pose=torch.zeros(1, 3),
                           leye_pose=torch.zeros(1, 3), reye_pose=torch.zeros(1, 3), left_hand_pose=torch.zeros(1, 45),
                           right_hand_pose=torch.zeros(1, 45), expression=torch.zeros(1, 5), transl=torch.zeros(1, 3))
        d.vs_template = out0.vertices; d.init_A = A0
        param = {'shape': betas, 'face_offset': face_offset, 'joint_offset': joint_offset, 'locator_offset': locator_offset,
                 'trans': trans.clone().requires_grad_(True), 'rhand_pose': hands[:, 1], 'lhand_pose': hands[:, 0],
                 'jaw_pose': jaw, 'expr': expr, 'body_pose': body_pose.clone().requires_grad_(True),
                 'root_pose': root_pose.clone().requires_grad_(True), 'leye_pose': eyes[:, 0], 'reye_pose': eyes[:, 1]}
        outs, As = [], []
        p_in = pts.clone().requires_grad_(True)
        gw = torch.randn(nfr, 300, 3, generator=gen)
        loss = 0
        for f in range(nfr):
            o = d.lbs_forward(p_in.reshape(1, -1, 3), param, idx=f)
            outs.append(o)
            loss = loss + (o * gw[f]).sum()
        loss.backward()
        w_pts = d.interpolate_weights(pts.reshape(1, -1, 3))
        can = d.apply_lbs_inverse(pts.reshape(1, -1, 3), A0, w_pts)
    # oracle check
    model = mt
    J0 = OL.joints_from_shape(model, betas, torch.zeros(1, 5))
    A0o = OL.pose_transforms(model, OL.full_pose(torch.zeros(1, 3), bp0, *[torch.zeros(1, 3)] * 3, torch.zeros(1, 45), torch.zeros(1, 45)), J0)
    assert (A0o - A0).abs().max() < 1e-6, (A0o - A0).abs().max()
    res = {'model.' + k: v for k, v in m.items() if k != 'posedirs'}   # posedirs never reaches the joint transforms
    for f in range(nfr):
        Jf = OL.joints_from_shape(model, betas, expr[f:f + 1], face_offset, joint_offset, locator_offset)
        Af = OL.pose_transforms(model, OL.full_pose(root_pose[f:f + 1], body_pose[f:f + 1], jaw[f:f + 1], eyes[f:f + 1, 0], eyes[f:f + 1, 1],
                                                    hands[f:f + 1, 0], hands[f:f + 1, 1]), Jf)
        o, idx, cano = OL.lbs_forward(pts, out0.vertices[0], mt['weights'], A0o[0], Af[0], trans[f])
        assert (o - outs[f].detach()).abs().max() < 2e-6, (f, (o - outs[f]).abs().max())
        As.append(Af[0])
    assert (cano - can[0]).abs().max() < 2e-6
    print('lbs: oracle == reference (A0, A, canonical, posed) for', nfr, 'frames')
    res.update({'betas': betas, 'expr': expr, 'body_pose': body_pose, 'root_pose': root_pose, 'jaw': jaw, 'hands': hands, 'eyes': eyes,
                'trans': trans, 'joint_offset': joint_offset, 'locator_offset': locator_offset, 'face_offset': face_offset,
                'pts': pts, 'A0': A0[0], 'A': torch.stack(As), 'tmpl': out0.vertices[0], 'w_pts': w_pts[0], 'canonical': can[0],
                'out': torch.stack([o.detach() for o in outs]), 'gout': gw, 'd_pts': p_in.grad, 'd_trans': param['trans'].grad,
                'd_body_pose': param['body_pose'].grad, 'd_root_pose': param['root_pose'].grad})
    np.savez_compressed(os.path.join(GOLD, 'lbs.npz'), **npy(res))


def gen_imgops():
    """reference auto_normals (render/mesh.py), prepare_shading_normal / image_loss (python twins, renderutils/ops.py use_python=True),
    ssim (ssim_loss.py) and compute_sdf_reg_loss (geometry/hmsdf.py) on seeded inputs, with gradients."""
    from oracle import image_ops as OI
    gen = torch.Generator().manual_seed(21)
    g8 = np.load(os.path.join(GOLD, 'mtets_gshell_n8.npz'))
    v = torch.from_numpy(g8['verts']).clone()
    f = torch.from_numpy(g8['faces']).clone()
    f = torch.cat([f, torch.tensor([[0, 0, 1]])])                 # one degenerate triangle
    res = {}
    with refharness.ref_ctx():
        from render import mesh as rmesh
        from render import renderutils as ru
        import ssim_loss as rssim
        vv = v.clone().requires_grad_(True)
        m = rmesh.auto_normals(rmesh.Mesh(vv, f))
        gn = torch.randn(m.v_nrm.shape, generator=gen)
        (m.v_nrm * gn).sum().backward()
        res.update({'an_v': v, 'an_f': f, 'an_out': m.v_nrm.detach(), 'an_g': gn, 'an_dv': vv.grad})
        # prepare_shading_normal, mirroring tests/test_bsdf.py:25-57 (RES=16 here)
        R = 16
        ins = {k: (torch.rand(1, R, R, 3, generator=gen) * 2 - 1) for k in ('pos', 'pert', 'snrm', 'stng', 'gnrm')}
        ins['view'] = torch.rand(1, 1, 1, 3, generator=gen) * 2 - 1
        for two_sided in (True, False):
            t = {k: x.clone().requires_grad_(True) for k, x in ins.items()}
            o = ru.prepare_shading_normal(t['pos'], t['view'], t['pert'], t['snrm'], t['stng'], t['gnrm'], two_sided_shading=two_sided,
                                          opengl=True, use_python=True)
            go = torch.randn(o.shape, generator=gen)
            (o * go).sum().backward()
            tag = 'psn2_' if two_sided else 'psn1_'
            res.update({tag + 'out': o.detach(), tag + 'g': go, **{tag + 'in_' + k: x for k, x in ins.items()},
                        **{tag + 'd_' + k: x.grad for k, x in t.items()}})
        # image loss (tonemapper 'none': python twin == CUDA kernel; log_srgb differs by the exposure factor, see oracle/image_ops.py)
        a = torch.rand(2, 12, 10, 3, generator=gen) * 2
        b = torch.rand(2, 12, 10, 3, generator=gen) * 2
        res.update({'il_a': a, 'il_b': b})
        for loss in ('l1', 'mse', 'smape', 'relmse'):
            aa, bb = a.clone().requires_grad_(True), b.clone().requires_grad_(True)
            l = ru.image_loss(aa, bb, loss=loss, tonemapper='none', use_python=True)
            l.backward()
            res.update({f'il_{loss}': l.detach(), f'il_{loss}_da': aa.grad, f'il_{loss}_db': bb.grad})
        # ssim
        x = torch.rand(2, 3, 40, 36, generator=gen)
        y = (x + 0.2 * torch.randn(x.shape, generator=gen)).clamp(0, 1)
        xx, yy = x.clone().requires_grad_(True), y.clone().requires_grad_(True)
        sv = rssim.ssim(xx, yy)
        sv.backward()
        res.update({'ssim_x': x, 'ssim_y': y, 'ssim': sv.detach(), 'ssim_dx': xx.grad, 'ssim_dy': yy.grad})
    # sdf_reg: geometry/hmsdf.py imports half the world at module level; exec only the function's source text is NOT allowed
    # (no reference source in the repo), so pin it through the module import with stubs
    try:
        for n in ['torchvision.models', 'torchvision.transforms', 'torchvision.transforms.functional', 'script.get_tet_smpl', 'kaolin.ops.mesh',
                  'PIL', 'PIL.Image']:
            refharness.stub(n)
        sys.modules['torchvision'].transforms = sys.modules['torchvision.transforms']
        sys.modules['torchvision'].models = sys.modules['torchvision.models']
        sys.modules['PIL'].Image = sys.modules['PIL.Image']
        sys.modules['script.get_tet_smpl'].get_tet_mesh = None
        ou = sys.modules['render.optixutils']
        for nm in ('OptiXContext', 'optix_build_bvh', 'optix_env_shade', 'bilateral_denoiser'):
            setattr(ou, nm, None)
        sys.modules['pytorch3d.io'].load_obj = None
        with refharness.ref_ctx():
            import importlib
            hm = importlib.import_module('geometry.hmsdf')
            sdf = torch.from_numpy(g8['in_sdf']).clone().requires_grad_(True)
            tets = torch.from_numpy(g8['tets'])
            e = tets[:, [0, 1, 0, 2, 0, 3, 1, 2, 1, 3, 2, 3]].reshape(-1, 2)
            e = torch.unique(torch.sort(e, dim=1)[0], dim=0)
            l = hm.compute_sdf_reg_loss(sdf[:, None], e)
            l.backward()
            res.update({'reg_sdf': sdf.detach(), 'reg_edges': e, 'reg': l.detach(), 'reg_dsdf': sdf.grad})
            assert abs(OI.sdf_reg_loss(sdf.detach(), e).item() - l.item()) < 1e-6
            print('imgops: sdf_reg pinned against geometry.hmsdf.compute_sdf_reg_loss')
    except Exception as ex:      # pragma: no cover
        print('imgops: geometry.hmsdf import failed, sdf_reg left to the restatement:', repr(ex)[:200])
    # oracle checks
    assert (OI.auto_normals(v, f) - res['an_out']).abs().max() < 1e-6
    for tag, ts in (('psn2_', True), ('psn1_', False)):
        o = OI.prepare_shading_normal(ins['pos'], ins['view'], ins['pert'], ins['snrm'], ins['stng'], ins['gnrm'], ts, True)
        assert (o - res[tag + 'out']).abs().max() < 1e-6
    for loss in ('l1', 'mse', 'smape', 'relmse'):
        assert abs(OI.image_loss(a, b, loss).item() - res[f'il_{loss}'].item()) < 1e-6
    assert abs(OI.ssim(x, y).item() - res['ssim'].item()) < 1e-6
    print('imgops: oracle == reference (auto_normals, prepare_shading_normal x2, image_loss x4, ssim)')
    np.savez_compressed(os.path.join(GOLD, 'imgops.npz'), **npy(res))


def gen_render():
    """reference render.render.render_mesh (render/render.py:347-451, incl. render_layer :213 and shade :42) driven by the oracle's
    nvdiffrast / tinycudann restatements (the real libraries are not installable here) on a 48x48 two-view frame -> all 12 buffers.
    Pins the composite / shade / buffer logic of the build's render.py against the reference's own render.py."""
    import types
    from oracle import raster as OR, texmlp as OT
    refharness.install()
    dr = sys.modules['nvdiffrast.torch']

    class Peeler:
        def __init__(self, ctx, pos, tri, res):
            self.a = (pos, tri, res)
        def __enter__(self):
            return self
        def __exit__(self, *a):
            return False
        def rasterize_next_layer(self):
            pos, tri, res = self.a
            return OR.rasterize(pos, tri.long(), res[0], res[1])
    dr.DepthPeeler = Peeler
    dr.interpolate = lambda attr, rast, tri, rast_db=None, diff_attrs=None: OR.interpolate(attr, rast, tri.long(), rast_db if diff_attrs is not None else None)
    dr.antialias = lambda color, rast, pos, tri: OR.antialias(color, rast, pos, tri.long())
    dr.texture = lambda tex, uv, filter_mode='linear', boundary_mode='clamp': OR.texture(tex, uv)
    tc = sys.modules['tinycudann']

    class Enc(torch.nn.Module):
        def __init__(self, n, cfg):
            super().__init__()
            self.n_output_dims = 10
            g = torch.Generator().manual_seed(3)
            self.params = torch.nn.Parameter((torch.rand(2 * OT.grid_layout()[1], generator=g) * 2 - 1) * 0.3)
        def forward(self, x):
            return OT.grid_encode(x, self.params)
    tc.Encoding = Enc
    tc.free_temporary_memory = lambda: None
    torch.nn.Module.cuda = lambda self, *a, **k: self
    g8 = np.load(os.path.join(GOLD, 'mtets_gshell_n8.npz'))
    res = 48
    with refharness.ref_ctx():
        from render import mesh as rmesh, render as rrender, mlptexture as rtex
        import functools
        _psn = rrender.ru.prepare_shading_normal     # the CUDA plugin cannot be built here: use the reference's own python twin
        rrender.ru.prepare_shading_normal = functools.partial(_psn, use_python=True)
        v = torch.from_numpy(g8['verts']).clone() * 1.1 + torch.tensor([0.0, -0.35, 0.0])
        f = torch.from_numpy(g8['faces']).clone()
        msdf = torch.from_numpy(g8['msdf']).clone()
        mn = torch.tensor([0, 0, 0, 0, 0.001, 0.0]); mx = torch.tensor([1, 1, 1, 0, 1.0, 1.0])
        torch.manual_seed(4)
        tex = rtex.MLPTexture3D((v.min(0).values, v.max(0).values), channels=6, min_max=[mn, mx])
        mat = {'kd_ks': tex, 'bsdf': 'pbr'}
        m = rmesh.auto_normals(rmesh.Mesh(v, f, material=mat))
        m_orig = rmesh.auto_normals(rmesh.Mesh(v * 0.97 + 0.01, f, material=mat))
        mv, mvp, campos = synth.camera(res, dist=3.0)
        ang = 0.4
        R = np.array([[np.cos(ang), 0, np.sin(ang), 0], [0, 1, 0, 0], [-np.sin(ang), 0, np.cos(ang), 0], [0, 0, 0, 1]], np.float32)
        mvps = torch.from_numpy(np.stack([mvp, mvp @ R]))
        camp = torch.from_numpy(np.stack([campos, (np.linalg.inv(mv @ R))[:3, 3]]).astype(np.float32))
        bg = torch.rand(2, res, res, 3, generator=torch.Generator().manual_seed(9))
        FL = types.SimpleNamespace(n_samples=1, decorrelated=False, denoiser_demodulate=False)
        torch.manual_seed(5)
        out = rrender.render_mesh(FL, 0, None, m, m_orig, mvps, camp, None, [res, res], spp=1, msaa=True, background=bg, use_uv=False,
                                  extra_dict={'msdf': msdf})
    resd = {'v': v, 'f': f, 'msdf': msdf, 'mvp': mvps, 'campos': camp, 'bg': bg, 'enc_seed': 3, 'enc_scale': 0.3,
            'w1': tex.net.net[0].weight.detach(), 'w2': tex.net.net[2].weight.detach(), 'w3': tex.net.net[4].weight.detach(), 'omin': mn, 'omax': mx}
    for k, t in out.items():
        resd['out.' + k] = t.detach()
        print('render golden', k, tuple(t.shape))
    np.savez_compressed(os.path.join(GOLD, 'render.npz'), **npy(resd))


def _patch_third_parties():
    """nvdiffrast / tinycudann / kaolin stand-ins for the reference modules: the oracle's restatements (the real libraries cannot be
    installed here); shared by gen_render and gen_tick_init"""
    from oracle import raster as OR, texmlp as OT
    refharness.install()
    dr = sys.modules['nvdiffrast.torch']

    class Peeler:
        def __init__(self, ctx, pos, tri, res):
            self.a = (pos, tri, res)
        def __enter__(self):
            return self
        def __exit__(self, *a):
            return False
        def rasterize_next_layer(self):
            pos, tri, res = self.a
            return OR.rasterize(pos, tri.long(), res[0], res[1])
    dr.DepthPeeler = Peeler
    dr.interpolate = lambda attr, rast, tri, rast_db=None, diff_attrs=None: OR.interpolate(attr, rast, tri.long(), rast_db if diff_attrs is not None else None)
    dr.antialias = lambda color, rast, pos, tri: OR.antialias(color, rast, pos, tri.long())
    dr.texture = lambda tex, uv, filter_mode='linear', boundary_mode='clamp': OR.texture(tex, uv)
    tc = sys.modules['tinycudann']

    class Enc(torch.nn.Module):
        def __init__(self, n, cfg):
            super().__init__()
            self.n_output_dims = 10
            g = torch.Generator().manual_seed(3)
            self.params = torch.nn.Parameter((torch.rand(2 * OT.grid_layout()[1], generator=g) * 2 - 1) * 0.3)
        def forward(self, x):
            return OT.grid_encode(x, self.params)
    tc.Encoding = Enc
    tc.free_temporary_memory = lambda: None
    torch.nn.Module.cuda = lambda self, *a, **k: self


def gen_tick_init():
    """The REFERENCE's HmSDFTetsGeometry.tick_init (geometry/hmsdf.py:810-915 through render_init :706 and getMesh_init :416) on a
    miniature scene: Kuhn n=12 grid, the SDF network pre-fitted to an ellipsoid, a 512-vertex synthetic body model, one 64 x 64 frame.
    The geometry object is built with object.__new__ (its __init__ needs pysdf / trimesh / tetgen / downloads); every method that
    runs is the reference's.  Stand-ins, all stated: nvdiffrast / tinycudann = the oracle restatements; kaolin sample_points = fixed
    pre-drawn surface samples; torchvision's pretrained MobileNetV2 = the same architecture with seeded random weights (the reference's
    own MobileNetPerceptualLoss class wraps it); loss_fn = the restatement of loss.cu (its CUDA plugin cannot be built here).
    Output: every loss term, total = reg + normal + msk (train.py:718) and its gradients; the oracle chain is asserted equal."""
    import types
    from oracle import tick as OTK, texmlp as OT, image_ops as OI, lbs as OL
    perceptual = _load_by_path('_d3h_perceptual_inputs', 'geometry/perceptual.py')
    _patch_third_parties()
    res, n = 64, 12
    gen = torch.Generator().manual_seed(41)
    verts_np, tets_np = synth.kuhn_grid(n)
    verts, tets = torch.from_numpy(verts_np), torch.from_numpy(tets_np)
    m = synth.make_body_model(n_verts=512, seed=0, n_shape=10, n_expr=5)
    mt = {k: torch.from_numpy(v) for k, v in m.items()}
    betas = torch.zeros(1, 10)
    body_pose = synth.poses(1, seed=1234) * 0.5
    root_pose, jaw, expr = torch.zeros(1, 3), torch.zeros(1, 3), torch.zeros(1, 5)
    trans0 = torch.tensor([[0.01, -0.02, 0.015]])
    for nme in ('torchvision.transforms', 'torchvision.transforms.functional', 'PIL', 'PIL.Image', 'tqdm'):
        try:
            __import__(nme)
        except Exception:
            refharness.stub(nme)
    if not hasattr(sys.modules['tqdm'], 'trange'):
        sys.modules['tqdm'].trange = range
    refharness.stub('script.get_tet_smpl', get_tet_mesh=None)
    refharness.stub('kaolin.ops.mesh')
    trunk_seed = 7
    tv = sys.modules['torchvision.models']
    tv.mobilenet_v2 = lambda pretrained=True: types.SimpleNamespace(features=perceptual.MobileNetPerceptualLoss(use_gpu=False, seed=trunk_seed).features)
    with refharness.ref_ctx():
        import geometry.hmsdf as rh
        from geometry.mlp import MLP
        from render import mlptexture as rtex
        import render.optixutils as rou
        from deform.smplx_exavatar.lbs import lbs as ref_lbs
        from deform.smplx_exavatar_deformer import SMPLX_Deformer
        rou.optix_build_bvh = lambda *a, **k: None
        rh.ou.optix_build_bvh = rou.optix_build_bvh
        import functools
        from render import render as rrender
        rrender.ru.prepare_shading_normal = functools.partial(rrender.ru.prepare_shading_normal, use_python=True)   # the reference's own python twin

        class Layer:                                   # as in gen_lbs: assembles full_pose (body_models.py:1225-1257), calls the reference lbs()
            lbs_weights = mt['weights']
            faces_tensor = None

            def forward(self, betas=None, global_orient=None, body_pose=None, jaw_pose=None, leye_pose=None, reye_pose=None,
                        left_hand_pose=None, right_hand_pose=None, expression=None, transl=None, face_offset=None,
                        joint_offset=None, locator_offset=None, pose2rot=True):
                fp = torch.cat([global_orient.reshape(-1, 1, 3), body_pose.reshape(-1, 21, 3), jaw_pose.reshape(-1, 1, 3),
                                leye_pose.reshape(-1, 1, 3), reye_pose.reshape(-1, 1, 3), left_hand_pose.reshape(-1, 15, 3),
                                right_hand_pose.reshape(-1, 15, 3)], dim=1).reshape(-1, 165)
                fp[:, 69:].zero_()
                comp = torch.cat([betas, expression], dim=-1)
                dirs = torch.cat([mt['shapedirs'], mt['expr_dirs']], dim=-1)
                vt = mt['v_template'] if face_offset is None else mt['v_template'] + face_offset
                v, j, A = ref_lbs(comp, fp, vt, dirs, mt['posedirs'], mt['J_regressor'], joint_offset, locator_offset, mt['parents'],
                                  mt['weights'], pose2rot=True)
                return types.SimpleNamespace(vertices=v + transl[:, None]), A
            __call__ = forward
        d = object.__new__(SMPLX_Deformer)
        d.layer = Layer(); d.lbs_weights = mt['weights']; d.k = 1; d.expr_param_dim = 5; d.shape_param_dim = 10
        bp0 = torch.zeros(1, 63); bp0[:, 2] = torch.pi / 36; bp0[:, 5] = -torch.pi / 36           # deformer.py:178-180
        z3, z45 = torch.zeros(1, 3), torch.zeros(1, 45)
        out0, A0 = d.layer(betas=betas, global_orient=z3, body_pose=bp0, jaw_pose=z3, leye_pose=z3, reye_pose=z3, left_hand_pose=z45,
                           right_hand_pose=z45, expression=torch.zeros(1, 5), transl=z3)
        d.vs_template, d.init_A = out0.vertices, A0

        # ---- the SDF network: reference MLP, default init, pre-fitted to an ellipsoid (hmsdf.py:254-271 loop) ----
        torch.manual_seed(0)
        net = MLP(skip_in=[3], n_freq=6, n_hidden=6, d_hidden=256)
        cen, rad = torch.tensor([0.0, -0.35, 0.0]), torch.tensor([0.5, 0.75, 0.42])
        sdf_gt = ((((verts - cen) / rad).norm(dim=-1) - 1.0) * 0.4).reshape(-1, 1)
        opt = torch.optim.Adam(net.parameters(), lr=1e-3)
        for _ in range(400):
            l = (net(verts) - sdf_gt).pow(2).mean()
            opt.zero_grad(); l.backward(); opt.step()
        print('tick_init: pre-fit loss', float(l))
        for p_ in net.parameters():
            p_.grad = None

        FL = types.SimpleNamespace(
            iter=2001, use_img_2nd_layer=False, use_depth=False, use_depth_2nd_layer=False, use_sdf_mlp=True, use_msdf_mlp=False,
            use_eikonal=True, eikonal_scale=None, sdf_regularizer=0.2, nonrigid_begin=20000, train_res=[res, res],
            visualize_watertight=False, n_samples=1, decorrelated=False, denoiser_demodulate=False,
            shape_param=betas, face_offset=None, joint_offset=None, locator_offset=None,
            trans_optim=trans0.clone().requires_grad_(True), rhand_pose_optim=torch.zeros(1, 45), lhand_pose_optim=torch.zeros(1, 45),
            jaw_pose_optim=jaw, expr_optim=expr, body_pose_optim=body_pose, root_pose_optim=root_pose, leye_pose_optim=z3, reye_pose_optim=z3)
        g = object.__new__(rh.HmSDFTetsGeometry)
        torch.nn.Module.__init__(g)
        g.FLAGS, g.grid_res, g.scale, g.batch_point_num = FL, 2 * n, 1.0, 100000
        g.gshell_tets = rh.GShell_Tets()
        g.smplx_deform = d
        g.optix_ctx = None
        g.verts, g.indices = verts, tets
        g.generate_edges()
        g.sdf_net = net
        g.sdf = None
        g.msdf = torch.nn.Parameter((torch.rand(verts.shape[0], generator=gen) - 0.15).clamp(-1, 1))       # mixed sign: an open surface
        g.deform = torch.nn.Parameter((torch.rand(verts.shape, generator=gen) * 2 - 1) * 0.3)
        g.mobileNet_perceptual_loss = rh.MobileNetPerceptualLoss(use_gpu=False)                            # the reference's own class
        mn = torch.tensor([0, 0, 0, 0, 0.001, 0.0]); mx = torch.tensor([1, 1, 1, 0, 1.0, 1.0])
        torch.manual_seed(4)
        tex = rtex.MLPTexture3D((verts.min(0).values, verts.max(0).values), channels=6, min_max=[mn, mx])
        mat = {'kd_ks': tex, 'bsdf': 'pbr'}
        mv, mvp, campos = synth.camera(res, dist=3.0)
        mvp_t, campos_t = torch.from_numpy(mvp)[None], torch.from_numpy(campos)[None]
        # targets: an ellipse mask displaced from the body, constant albedo, smooth unit normals inside the mask
        yy, xx = torch.meshgrid(torch.arange(res, dtype=torch.float32), torch.arange(res, dtype=torch.float32), indexing='ij')
        msk = ((((xx - 33.5) / 11.0) ** 2 + ((yy - 30.0) / 16.5) ** 2) < 1).float()[None, ..., None]
        nx, ny = (xx - 33.5) / 11.0, -(yy - 30.0) / 16.5
        nz = (1 - (nx ** 2 + ny ** 2)).clamp(min=0.05).sqrt()
        nrm = torch.nn.functional.normalize(torch.stack([nx, ny, nz], -1), dim=-1)[None] * msk
        all_img = torch.cat([torch.tensor([0.55, 0.45, 0.40]).expand(1, res, res, 3) * msk, msk], -1)
        bg = torch.rand(1, res, res, 3, generator=gen)
        target = {'idx': [0], 'mvp': mvp_t, 'campos': campos_t, 'resolution': [res, res], 'spp': 1, 'background': bg, 'all_img': all_img,
                  'all_normal': nrm}
        # fixed surface samples for the eikonal term: drawn on the posed mesh of this very state (one dry run of getMesh_init)
        with torch.no_grad():
            dry = g.getMesh_init(mat, target=target)
            pts = OTK.surface_samples(dry['deform_imesh'].v_pos, dry['deform_imesh'].t_pos_idx, 2000, generator=gen)
        sys.modules['kaolin'].ops.mesh.sample_points = lambda v, f, k: (pts[None], None)
        rh.kaolin = sys.modules['kaolin']
        loss_fn = lambda img, ref: OI.image_loss(img, ref, 'l1', 'log_srgb')
        it = 120
        draws_seed = 5
        torch.manual_seed(draws_seed)
        r = g.tick_init(None, target, None, mat, loss_fn, it, None)
        total = r['reg_loss'] + r['normal_loss'] + r['msk_loss']                                             # train.py:718
        total.backward()
        md = g.last_mesh if hasattr(g, 'last_mesh') else None
    out = {'verts': verts, 'indices': tets, 'deform': g.deform.detach(), 'msdf': g.msdf.detach(), 'grid_res': 2 * n, 'res': res,
           'iteration': it, 'n_iter': FL.iter, 'sdf_regularizer': 0.2, 'draws_seed': draws_seed, 'trunk_seed': trunk_seed,
           'betas': betas, 'expr': expr, 'body_pose': body_pose, 'root_pose': root_pose, 'jaw': jaw, 'trans': trans0,
           'tmpl': d.vs_template[0], 'A0': A0[0], 'mvp': mvp_t, 'campos': campos_t, 'bg': bg, 'all_img': all_img, 'all_normal': nrm,
           'sampled_pts': pts, 'enc_seed': 3, 'enc_scale': 0.3, 'omin': mn, 'omax': mx,
           'w1': tex.net.net[0].weight.detach(), 'w2': tex.net.net[2].weight.detach(), 'w3': tex.net.net[4].weight.detach(),
           'n_mesh_verts': dry['deform_imesh'].v_pos.shape[0], 'n_mesh_faces': dry['deform_imesh'].t_pos_idx.shape[0]}
    for k, v in m.items():
        if k != 'posedirs':
            out['model.' + k] = v
    for k, v in net.state_dict().items():
        out['sd.' + k] = v
    for k, v in r.items():
        out['loss.' + k] = v.detach()
    out['loss.total'] = total.detach()
    for k, p_ in net.named_parameters():
        out['grad.sd.' + k] = p_.grad
    out['grad.deform'], out['grad.msdf'], out['grad.trans'] = g.deform.grad, g.msdf.grad, FL.trans_optim.grad
    out['grad.table'] = tex.encoder.params.grad
    for i, k in zip((0, 2, 4), ('w1', 'w2', 'w3')):
        out['grad.' + k] = tex.net.net[i].weight.grad
    print('tick_init:', {k: float(v) for k, v in r.items()}, 'mesh', out['n_mesh_verts'], out['n_mesh_faces'])

    # ---- the oracle chain on the same inputs must give the same numbers ----
    st = OTK.state_from_golden(npy(out), perceptual.MobileNetPerceptualLoss)
    torch.manual_seed(draws_seed)
    ro = OTK.tick_init(st, buffers=None)
    ro['total'].backward()
    for k in ('img_loss', 'msk_loss', 'eik_loss', 'sdf_reg_loss', 'reg_loss', 'normal_loss'):
        a, b = float(ro[k]), float(r[k])
        assert abs(a - b) <= 1e-5 * max(1e-3, abs(b)), (k, a, b)
    rel = lambda a, b: float((a - b).abs().max() / (b.abs().max() + 1e-20))
    for k, p_ in net.named_parameters():
        assert rel(st['sd'][k].grad, p_.grad) < 1e-3, (k, rel(st['sd'][k].grad, p_.grad))
    assert rel(st['deform'].grad, g.deform.grad) < 1e-3 and rel(st['msdf'].grad, g.msdf.grad) < 1e-3
    assert rel(st['trans'].grad, FL.trans_optim.grad) < 1e-3
    assert rel(st['material']['table'].grad, tex.encoder.params.grad) < 1e-3
    print('tick_init: oracle chain == reference tick_init (6 loss terms, all parameter gradients)')
    np.savez_compressed(os.path.join(GOLD, 'tick_init.npz'), **npy(out))



def gen_lpips():
    """the reference's vendored LPIPS (third_parties/lpips/lpips.py:19-145) with its calibrated linear layers (weights/v0.1/{alex,vgg}.pth)
    on a seeded RANDOM trunk (torchvision and its ImageNet weights are not available: `torchvision.models.alexnet / vgg16` are
    stand-ins that build the same layer stack through the build's own trunk builder, so both sides hold identical weights).
    Output: the distance and its gradient w.r.t. the first image for both trunks; the linear weights are stored as data."""
    import types
    blp = _load_pkg('_d3h_lpips_inputs', 'lpips/__init__.py')
    refharness.install()
    tv = sys.modules['torchvision.models']
    seed = 11

    def features_of(layers, slices):
        tr = blp._Trunk(layers, slices, seed=seed)
        feats = {}
        for k in range(tr.N_slices):
            for name, mod in getattr(tr, f'slice{k + 1}').named_children():
                feats[int(name)] = mod
        return torch.nn.Sequential(*[feats[i] for i in range(len(feats))])
    tv.alexnet = lambda weights=None, **k: types.SimpleNamespace(features=features_of(blp._ALEX, blp._ALEX_SLICES))
    tv.vgg16 = lambda weights=None, **k: types.SimpleNamespace(features=features_of(blp._vgg_layers(), blp._VGG_SLICES))
    refharness.stub('torchvision').models = tv
    sys.path.insert(0, os.path.join(refharness.REF, 'third_parties'))
    for n in ('skimage', 'skimage.measure', 'skimage.color', 'scipy.ndimage', 'tqdm', 'IPython'):
        try:
            __import__(n)
        except Exception:
            refharness.stub(n)
    out = {'trunk_seed': seed}
    gen = torch.Generator().manual_seed(5)
    a, b = torch.rand(2, 3, 64, 64, generator=gen), torch.rand(2, 3, 64, 64, generator=gen)
    out['in0'], out['in1'] = a, b
    with refharness.ref_ctx():
        import lpips as rl
        assert 'reference' in rl.__file__
        for net in ('alex', 'vgg'):
            mref = rl.LPIPS(net=net, pnet_rand=True, verbose=False)              # pretrained=True: loads the vendored linear layers
            x = a.clone().requires_grad_(True)
            val, per = mref(x, b, retPerLayer=True)
            val.sum().backward()
            out[f'{net}.val'], out[f'{net}.d_in0'] = val.detach(), x.grad
            for k, r in enumerate(per):
                out[f'{net}.layer{k}'] = r.detach()
            for k in range(5):
                out[f'{net}.lin{k}'] = mref.state_dict()[f'lin{k}.model.1.weight']
            # the build's module with the same trunk seed and the same linear weights must agree
            mb = blp.LPIPS(net=net, pretrained=False, trunk_seed=seed)
            mb.load_state_dict({f'lin{k}.model.1.weight': out[f'{net}.lin{k}'] for k in range(5)}, strict=False)
            assert set(mb.state_dict().keys()) == set(mref.state_dict().keys()), set(mb.state_dict().keys()) ^ set(mref.state_dict().keys())
            mb.load_state_dict(mref.state_dict())                              # a checkpoint of the reference module loads unchanged
            y = a.clone().requires_grad_(True)
            vb = mb(y, b)
            vb.sum().backward()
            assert (vb - val).abs().max() < 1e-6 * max(1.0, float(val.abs().max())), (net, float((vb - val).abs().max()))
            assert (y.grad - x.grad).abs().max() < 1e-5 * float(x.grad.abs().max())
            print('lpips', net, 'value', val.reshape(-1).tolist(), '== build')
    np.savez_compressed(os.path.join(GOLD, 'lpips.npz'), **npy(out))


def _icosphere(sub):
    """closed manifold triangle mesh: subdivided octahedron projected on the unit sphere"""
    v = [(1, 0, 0), (-1, 0, 0), (0, 1, 0), (0, -1, 0), (0, 0, 1), (0, 0, -1)]
    f = [(0, 2, 4), (2, 1, 4), (1, 3, 4), (3, 0, 4), (2, 0, 5), (1, 2, 5), (3, 1, 5), (0, 3, 5)]
    v = [np.array(p, np.float64) for p in v]
    for _ in range(sub):
        mid, nf = {}, []

        def m(a, b):
            k = (min(a, b), max(a, b))
            if k not in mid:
                p = v[a] + v[b]
                v.append(p / np.linalg.norm(p))
                mid[k] = len(v) - 1
            return mid[k]
        for a, b, c in f:
            ab, bc, ca = m(a, b), m(b, c), m(c, a)
            nf += [(a, ab, ca), (b, bc, ab), (c, ca, bc), (ab, bc, ca)]
        f = nf
    return np.array(v, np.float32), np.array(f, np.int64)


def gen_seq():
    """seq-stage geometry terms: collision / uniform Laplacian / normal consistency / connected faces / MLP_deform"""
    from oracle import seq_ops as OS
    bv, bf = _icosphere(2)                      # body: 66 verts / 128 faces
    cv, cf = _icosphere(2)
    g = torch.Generator().manual_seed(21)
    body_v = torch.from_numpy(bv) * torch.tensor([0.5, 0.8, 0.4]) + 0.01 * torch.randn(bv.shape, generator=g)
    cloth_v = torch.from_numpy(cv) * torch.tensor([0.52, 0.6, 0.43]) + torch.tensor([0.0, 0.1, 0.0]) + 0.01 * torch.randn(cv.shape, generator=g)
    body_f, cloth_f = torch.from_numpy(bf), torch.from_numpy(cf)
    all_v = torch.cat([body_v, cloth_v])
    all_f = torch.cat([body_f, cloth_f + body_v.shape[0]])
    out = {'body_v': body_v, 'cloth_v': cloth_v, 'body_f': body_f, 'cloth_f': cloth_f, 'all_v': all_v, 'all_f': all_f}
    with refharness.ref_ctx():
        # geometry/hmsdf.py:10-45 imports: stub what the container lacks (none of it is on the collision_loss path)
        for n in ('torchvision.transforms', 'torchvision.transforms.functional', 'PIL', 'PIL.Image', 'tqdm'):
            try:
                __import__(n)
            except Exception:
                refharness.stub(n)
        if not hasattr(sys.modules['tqdm'], 'trange'):
            sys.modules['tqdm'].trange = range
        refharness.stub('script.get_tet_smpl', get_tet_mesh=None)
        import geometry.hmsdf as rh
        import render.mesh as rmesh
        import lap_loss as rlap
        from geometry.mlp import MLP_deform
        c = cloth_v.clone().requires_grad_(True)
        b = body_v.clone().requires_grad_(True)
        l = rh.collision_loss(c, b, body_f, push_eps=0.05)      # larger eps than the default so many vertices are active
        l.backward()
        out.update({'colli': l.detach(), 'colli_dcloth': c.grad, 'colli_dbody': b.grad, 'colli_eps': 0.05})
        l0 = rh.collision_loss(cloth_v, body_v, body_f)
        out['colli_default'] = l0
        conn, e_all = rmesh.find_connected_faces(all_f)
        out['connected_faces'] = conn
        out['edges_unique'] = rmesh.find_edges(all_f)
        v = all_v.clone().requires_grad_(True)
        m = rmesh.Mesh(v, all_f, connected_faces=conn)
        out['mesh_edges'] = m.edges
        ll = rlap.body_laplacian_loss(m)
        ll.backward()
        out.update({'lap': ll.detach(), 'lap_dv': v.grad.clone()})
        v.grad = None
        nl = rlap.body_normal_loss(m)
        nl.backward()
        out.update({'ncons': nl.detach(), 'ncons_dv': v.grad.clone()})
        torch.manual_seed(6)
        net = MLP_deform(skip_in=[3], n_freq=8, n_hidden=6, d_hidden=256, d_out=3)
        code = 0.1 * torch.randn(1, 1, 136, generator=g)
        code.requires_grad_(True)
        x = all_v[:96].reshape(1, -1, 3)
        y = net(x, code)
        wgt = torch.randn(y.shape, generator=g)
        (y * wgt).sum().backward()
        out.update({'nr_x': x, 'nr_code': code.detach(), 'nr_y': y.detach(), 'nr_w': wgt, 'nr_dcode': code.grad})
        sd = {k: p.detach().clone() for k, p in net.state_dict().items()}
        for k, p in net.named_parameters():
            out['nr_grad.' + k] = p.grad
        for k, p in sd.items():
            out['nr_sd.' + k] = p
        skip = list(net.skip_count)
    # ---- oracle check ----
    assert abs(OS.collision_loss(cloth_v, body_v, body_f, 0.05).item() - out['colli'].item()) < 1e-7 * max(1, abs(out['colli'].item()))
    oc, _ = OS.find_connected_faces(all_f)
    assert torch.equal(oc, out['connected_faces']), 'connected faces differ'
    assert torch.equal(OS.find_edges(all_f), out['edges_unique'])
    assert abs(OS.laplacian_uniform_loss(all_v, out['mesh_edges']).item() - out['lap'].item()) < 1e-6 * out['lap'].item()
    assert abs(OS.normal_consistency_loss(all_v, all_f, oc).item() - out['ncons'].item()) < 1e-5 * out['ncons'].item()
    yo = OS.mlp_deform_forward(x, out['nr_code'], sd, n_freq=8, skip_layers=tuple(skip))
    assert (yo - out['nr_y']).abs().max() < 1e-6, (yo - out['nr_y']).abs().max()
    out['nr_skip_layers'] = np.array(skip)
    print('seq: colli', float(out['colli']), 'lap', float(out['lap']), 'ncons', float(out['ncons']), 'pairs', tuple(oc.shape), 'skip', skip)
    np.savez_compressed(os.path.join(GOLD, 'seq.npz'), **npy(out))


def gen_data_edges():
    """reference Dataset_split (dataset/dataset_split.py): the camera block of __init__ (:164-204, re-run on its own lines through a
    bare instance) and the REAL __getitem__ (:206-283) with imageio / cv2 replaced by in-memory arrays (resize = identity at the
    target size, BGR<->RGB as a channel flip) -> the target dict the tick_* functions consume."""
    import importlib, types
    refharness.install()
    ds = importlib.import_module('dataset.dataset_split')
    import imageio, cv2
    H, W = 12, 10
    rng = np.random.default_rng(5)
    K = np.array([[1201.0, 0, 541.0], [0, 1199.0, 539.0], [0, 0, 1]])
    w2c = np.eye(4, dtype=np.float32); w2c[:3, 3] = [0.1, -0.2, 2.5]
    # ---- camera: dataset_split.py:164-204 ----
    Kt, w2ct = torch.from_numpy(K), torch.from_numpy(w2c).float()
    height, width = 1080 // 2, 1080 // 2
    fx, fy, cx, cy = Kt[0, 0] // 2, Kt[1, 1] // 2, Kt[0, 2] // 2, Kt[1, 2] // 2
    proj = ds.get_ndc_matrix_from_ss(height, width, fx, fy, cx, cy)
    flip = torch.tensor([[1, 0, 0, 0], [0, -1, 0, 0], [0, 0, -1, 0], [0, 0, 0, 1]], dtype=torch.float)
    mv = flip @ w2ct
    campos = torch.linalg.inv(mv)[:3, 3]
    mvp = proj @ mv
    # ---- __getitem__ on in-memory "files" ----
    rgb = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    nrm = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    msk = (rng.random((H, W)) > 0.4).astype(np.uint8) * 255
    cloth = (msk * (rng.random((H, W)) > 0.5)).astype(np.uint8)
    body = (msk * (cloth == 0)).astype(np.uint8)
    files = {'rgb': rgb, 'msk': msk, 'cloth': cloth, 'body': body}
    imageio.imread = lambda p: files[p].copy()
    cv2.resize = lambda a, sz: a
    cv2.IMREAD_COLOR, cv2.COLOR_BGR2RGB = 1, 4
    cv2.imread = lambda p, flag: nrm[..., ::-1].copy()           # cv2 decodes to BGR
    cv2.cvtColor = lambda a, code: a[..., ::-1].copy()
    o = object.__new__(ds.Dataset_split)
    o.key_frame, o.n_images, o.examples = [0], 1, None
    o.FLAGS = types.SimpleNamespace(train_res=[H, W], spp=1)
    o.img_lists, o.msk_lists, o.cloth_msk_lists, o.body_msk_lists, o.normal_lists = ['rgb'], ['msk'], ['cloth'], ['body'], ['nrm']
    o.mv, o.mvp, o.campos = mv, mvp, campos
    z = torch.zeros(1, 3)
    o.smplx_params = {k: z for k in ('trans', 'rhand_pose', 'jaw_pose', 'expr', 'body_pose', 'root_pose', 'lhand_pose', 'leye_pose')}
    with refharness.ref_ctx():
        t = o.__getitem__(0)
    out = {'proj': proj, 'mv': mv, 'mvp': mvp, 'campos': campos, 'K': K, 'w2c': w2c, 'rgb': rgb, 'nrm': nrm, 'msk': msk, 'cloth': cloth,
           'body': body}
    for k in ('all_img', 'cloth_img', 'body_img', 'all_normal', 'body_normal', 'cloth_normal', 'all_msk', 'cloth_msk', 'body_msk', 'mv', 'mvp',
              'campos'):
        out['t.' + k] = t[k]
    print('data_edges: all_img', tuple(t['all_img'].shape), t['all_img'].dtype, 'normal', t['all_normal'].dtype)
    np.savez_compressed(os.path.join(GOLD, 'data_edges.npz'), **npy(out))


ALL = {'tick_init': gen_tick_init, 'lpips': gen_lpips, 'sdf_mlp': gen_sdf_mlp, 'mtets': gen_mtets, 'lbs': gen_lbs, 'imgops': gen_imgops, 'render': gen_render, 'seq': gen_seq, 'data_edges': gen_data_edges}

if __name__ == '__main__':
    names = sys.argv[1:] or list(ALL)
    if len(names) == 1:
        ALL[names[0]]()
    else:
        # one process per generator: each one patches module globals (stub third parties, torch.nn.Module.cuda) for its own needs
        import subprocess
        for nme in names:
            subprocess.check_call([sys.executable, os.path.abspath(__file__), nme])
