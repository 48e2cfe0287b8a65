"""Drive the REFERENCE's own train.py on top of this build -- the drop-in claim of INTEGRATION.md section 2, executed.

    PYTHONPATH=<repo>/d3human-code_amd:<reference root>  python tools/run_reference_train.py [--stage init|split|seq|all] [--iters 2]
                                                                                          [--emulator] [--res 32] [--grid 6]

What runs is the reference's `train.optimize_mesh_init` (train.py:544-832: its optimiser groups, LambdaLR schedulers, DataLoader loop,
prepare_batch_init, tick_init call, total = reg + normal + msk, backward, encoder-gradient scaling, Adam steps, clamp, stream sync),
`train.optimize_mesh_split` (train.py:839-1243: prepare_batch_split, validate_itr_all at iteration 0, tick_split for the garment and the
body, the total of :1087, both optimisers) and `train.optimize_mesh_seq` (train.py:1246-1525: tick_seq, the total of :1412-1421,
validate_all_mesh + the .ply / .npz outputs at the end).  optimize_mesh_seq loops a hard-coded 1000 (or 300) times; the harness
shortens that by giving the train MODULE a `range` that yields the first `--iters` indices and the last one (the function, hence the
file, is untouched; every statement of the loop body, the final validation and the delta / visible-triangle dump run).  All with
`geometry`, `render`, `deform`, `nvdiffrast`, `tinycudann`, `kaolin`, `pytorch3d`, `ssim_loss`, `lap_loss` resolving to this build and
`dataset` resolving to this build and everything else (`render.{material,texture,light}`, `denoiser`, `script`) to the reference's files.  Inputs the
repository does not ship are synthetic: the sequence (the build's Dataset_split over an in-memory frame), the SMPL-X model (d3h.synth), the tet grid, the merged body + garment mesh of the seq stage (an ellipsoid and an open tube, labels
prepared as train.py:1885-1911 does).  Third-party modules that train.py imports at the top but these stages never call
(xatlas, cv2, openmesh, tensorboardX, imageio, open3d, pymeshlab, pysdf, trimesh, torchvision) are stubbed when absent.

--emulator: no GPU in the dev container -- the kernels run through the test-only host emulation and the reference's hard-coded
device='cuda' is rewritten to 'cpu' (tools/refharness.CudaToCpu).  On a GPU box neither is needed.
"""
import argparse
import os
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _stub_missing(names):
    for name in names:
        try:
            __import__(name)
            continue
        except Exception:
            pass
        parts = name.split('.')
        for i in range(1, len(parts) + 1):
            n = '.'.join(parts[:i])
            if n not in sys.modules:
                m = types.ModuleType(n)
                m.__path__ = []
                sys.modules[n] = m
                if i > 1:
                    setattr(sys.modules['.'.join(parts[:i - 1])], parts[i - 1], m)


class _Writer:                                    # tensorboardX.SummaryWriter stand-in
    def __init__(self, *a, **k):
        self.scalars = []

    def add_scalar(self, tag, value, step):
        self.scalars.append((tag, float(value), step))

    def close(self):
        pass


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--iters', type=int, default=2)
    ap.add_argument('--stage', default='init', choices=('init', 'split', 'seq', 'all'))
    ap.add_argument('--res', type=int, default=32)
    ap.add_argument('--grid', type=int, default=6)
    ap.add_argument('--emulator', action='store_true')
    ap.add_argument("--eik", type=int, default=256)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    import numpy as np
    import torch
    _stub_missing(['xatlas', 'cv2', 'openmesh', 'tensorboardX', 'imageio', 'open3d', 'pymeshlab', 'pysdf', 'trimesh', 'torchvision',
                   'torchvision.models', 'torchvision.transforms', 'PIL', 'PIL.Image', 'tqdm'])
    if not hasattr(sys.modules['tensorboardX'], 'SummaryWriter'):
        sys.modules['tensorboardX'].SummaryWriter = _Writer
    ctx = None
    if a.emulator:
        from d3h import _lib as L
        L._use_emulator_for_tests(os.path.join(ROOT, 'tests', 'emul', 'libd3h_emul.so'))
        sys.path.insert(0, os.path.join(ROOT, 'tools'))
        import refharness
        ctx = refharness.CudaToCpu()
        ctx.__enter__()
        torch.nn.Module.cuda = lambda self, *x, **k: self
        torch.cuda.current_stream = lambda *x, **k: types.SimpleNamespace(synchronize=lambda: None)
    dev = 'cpu' if a.emulator else 'cuda'

    import train                                                     # the reference's train.py
    import geometry.hmsdf, render.render, render.util, render.material, dataset.dataset_split   # noqa: F401,E401
    where = {m: os.path.relpath(sys.modules[m].__file__, ROOT) if sys.modules[m].__file__.startswith(ROOT) else sys.modules[m].__file__
             for m in ('train', 'geometry.hmsdf', 'render.render', 'render.util', 'render.material', 'dataset.dataset_split', 'nvdiffrast.torch')}
    print("resolved:", where, flush=True)
    assert where['geometry.hmsdf'].startswith('d3human-code_amd') and where['render.render'].startswith('d3human-code_amd')
    assert 'd3human-code_amd' not in where['train'] and 'd3human-code_amd' not in where['render.material']
    assert where['dataset.dataset_split'].startswith('d3human-code_amd')              # train.py:25 `from dataset.dataset_split import Dataset_split`

    from d3h import synth
    from d3h.scene import make_flags
    import nvdiffrast.torch as dr
    F = make_flags(res=a.res, grid_n=a.grid, n_frames=1, device=dev, prefit_steps=200, body_verts=512,
                   sdf_fn=lambda x: (((x - torch.tensor([0.0, -0.4, 0.0], device=x.device)) / torch.tensor([0.55, 0.8, 0.45], device=x.device)).norm(dim=-1) - 1.0) * 0.4)
    # the fields train.py's __main__ sets before it builds the geometry (train.py:1566-1640, 1700-1726)
    F.iter, F.batch, F.loss, F.background, F.bsdf = a.iters, 1, 'logl1', 'black', 'pbr'
    F.learning_rate = [0.03, 0.005]            # configs/f3c.json; train.py:569 indexes it with pass_idx = 0 -> 0.03 for positions AND material
    F.local_rank, F.display_interval, F.save_interval, F.save_checkpoint_interval = 0, 0, 0, 10 ** 9
    F.clip_max_norm, F.normal_only, F.no_perturbed_nrm, F.nonrigid_begin = 0.0, False, False, 20000
    F.use_img_2nd_layer = F.use_depth = F.use_depth_2nd_layer = False
    F.eikonal_samples = a.eik
    F.prefit_with_library_path = a.emulator      # start-up pre-fit through library GEMMs: the emulated MFMA sweeps would take minutes
    z = lambda n: torch.zeros(1, n, device=dev)
    F.trans, F.rhand_pose, F.jaw_pose, F.expr = F.trans_optim, z(45), F.jaw_pose_optim, F.expr_optim
    F.body_pose, F.root_pose, F.lhand_pose, F.leye_pose, F.reye_pose = F.body_pose_optim, F.root_pose_optim, z(45), z(3), z(3)
    F.rhand_pose_optim, F.lhand_pose_optim, F.leye_pose_optim, F.reye_pose_optim = z(45), z(45), z(3), z(3)
    F.out_dir = a.out
    stages = ('init', 'split', 'seq') if a.stage == 'all' else (a.stage,)
    # train.py:1555-1617: the fields the split / seq stages read
    F.use_mesh_msdf_reg, F.msdf_reg_open_scale, F.msdf_reg_close_scale = True, 1e-6, 3e-6
    F.lambda_kd, F.lambda_ks, F.lambda_nrm, F.lambda_chroma, F.lambda_diffuse, F.lambda_specular = 0.1, 0.05, 0.025, 0.0, 0.15, 0.0025
    F.use_nonrigid_deform, F.deform_checkpoint, F.sdf_deform_pretrain_steps = True, None, 20
    F.texture_res = [a.res, a.res]
    F.seq_epoch = 2

    geometry_obj = train.HmSDFTetsGeometry(2 * a.grid, 1.0, F)                      # == geometry.hmsdf of this build
    mat = train.initial_guess_material(geometry_obj, True, F, None)
    mat['no_perturbed_nrm'] = True
    mv, mvp, campos = synth.camera(a.res, dist=3.0)
    H = W = a.res
    yy, xx = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing='ij')
    msk = ((((xx - 0.5 * W) / (0.2 * W)) ** 2 + ((yy - 0.45 * H) / (0.3 * H)) ** 2) < 1).float()[None, ..., None]
    img = torch.cat([torch.tensor([0.55, 0.45, 0.40]).expand(1, H, W, 3) * msk, msk], -1)
    nrm = torch.nn.functional.normalize(torch.stack([(xx - 0.5 * W) / W, -(yy - 0.45 * H) / H, torch.ones_like(xx)], -1), dim=-1)[None] * msk

    # the build's Dataset_split (what train.py:25 imports) over an in-memory sequence: one synthetic frame repeated, its images /
    # masks / normal map as uint8 arrays (what the PNG decoder would hand it), the camera as a calibration at twice the training resolution
    from dataset.dataset_split import MemorySource, SMPLX_KEYS, _SMPLX_WIDTH
    u8 = lambda t: (t.clamp(0, 1) * 255).round().to(torch.uint8).numpy()
    mask_u8 = u8(msk[0, ..., 0])
    rgb_u8 = u8(torch.tensor([0.55, 0.45, 0.40]).expand(H, W, 3) * msk[0])
    nrm_u8 = u8((nrm[0] + 1) / 2)
    n_fr = a.iters + 2
    smplx = {k: np.zeros((n_fr, _SMPLX_WIDTH[k]), np.float32) for k in SMPLX_KEYS}
    smplx.update(face_offset=np.zeros((1, 8, 3), np.float32), joint_offset=np.zeros((1, 55, 3), np.float32),
                 locator_offset=np.zeros((1, 55, 3), np.float32), shape_param=np.zeros((1, 100), np.float32))
    K2 = np.array([[2 * 1.2 * W, 0, W], [0, 2 * 1.2 * H, H], [0, 0, 1]], np.float64)            # halved by the dataset (dataset_split.py:170-179)
    w2c = np.eye(4, dtype=np.float32); w2c[1, 1] = w2c[2, 2] = -1; w2c[:3, 3] = (0, -0.43, 3.0)   # d3h.synth.camera's pose
    src = MemorySource([(rgb_u8, mask_u8, mask_u8, mask_u8, nrm_u8)] * n_fr, (0, n_fr - 1), smplx,
                       {'intrinsic': K2, 'extrinsic': w2c, 'height': 2 * H, 'width': 2 * W})
    F.train_res, F.spp = [H, W], 1
    data = dataset.dataset_split.Dataset_split(src, F, examples=a.iters + 1)
    # every frame of this sequence is posed with the FLAGS tensors (train.py copies the optimised rows there); frame index 0 always
    data.key_frame = [0] * len(data.key_frame)
    out = a.out or os.path.join('/tmp', 'd3h_train_drive')
    os.makedirs(out, exist_ok=True)
    glctx = dr.RasterizeGLContext()
    finite = lambda: all(torch.isfinite(p).all() for p in geometry_obj.parameters())
    if 'init' in stages:
        before = [p.detach().clone() for p in geometry_obj.sdf_net.parameters()]
        ret = train.optimize_mesh_init(None, glctx, geometry_obj, mat, None, data, data, F, warmup_iter=1, log_interval=1,
                                       pass_idx=0, pass_name='init', optimize_light=False, optimize_geometry=True, visualize=False,
                                       save_path=out)
        moved = max(float((p.detach() - b).abs().max()) for p, b in zip(geometry_obj.sdf_net.parameters(), before))
        print('optimize_mesh_init returned', type(ret).__name__, '; SDF weights moved by', moved)
        assert moved > 0, 'the optimiser of train.py did not update the SDF network'
        assert finite()
    if 'split' in stages:
        # the 448-crop of the perceptual normal loss (hmsdf.py:1072) is only taken when a network is plugged in; none offline
        before = (geometry_obj.deform.detach().clone(), geometry_obj.msdf.detach().clone(), mat['kd_ks'].encoder.params.detach().clone())
        sd0 = [p.detach().clone() for p in geometry_obj.sdf_net.parameters()]
        ret = train.optimize_mesh_split(None, glctx, geometry_obj, mat, None, data, data, F, warmup_iter=1, log_interval=1, pass_idx=0,
                                        pass_name='split', optimize_light=False, optimize_geometry=True, visualize=True, save_path=out)
        moved = [float((x.detach() - y).abs().max()) for x, y in zip((geometry_obj.deform, geometry_obj.msdf, mat['kd_ks'].encoder.params), before)]
        print('optimize_mesh_split returned', type(ret).__name__, '; deform / msdf / texture table moved by', moved)
        assert all(m > 0 for m in moved), 'the optimisers of train.py did not update deform / msdf / material'
        # the split stage's optimiser holds no SDF-network parameter (train.py:887-902): it must not have moved
        assert all(torch.equal(p.detach(), q) for p, q in zip(geometry_obj.sdf_net.parameters(), sd0))
        assert finite()
    if 'seq' in stages:
        import builtins
        from render import mesh as rmesh
        bv, bf = synth.icosphere(2)
        cv, cf = synth.tube(12, 4)
        body_v = torch.from_numpy(bv) * torch.tensor([0.45, 0.7, 0.38]) + torch.tensor([0.0, -0.4, 0.0])
        cloth_v = torch.from_numpy(cv) * torch.tensor([0.56, 0.2, 0.46]) + torch.tensor([0.0, -0.42, 0.15])
        v = torch.cat([body_v, cloth_v]).float().to(dev).contiguous()
        f = torch.cat([torch.from_numpy(bf), torch.from_numpy(cf) + body_v.shape[0]]).long().to(dev).contiguous()
        face_labels = torch.cat([torch.zeros(bf.shape[0], dtype=torch.long), torch.ones(cf.shape[0], dtype=torch.long)]).to(dev)
        # train.py:1880-1911
        F.v, F.f, F.face_labels = v, f, face_labels
        F.body_f, F.cloth_f = f[face_labels == 0], f[face_labels == 1]
        num_labels = int(face_labels.max().item()) + 1
        counts = torch.bincount(f.reshape(-1) * num_labels + face_labels.unsqueeze(1).expand(-1, 3).reshape(-1), minlength=v.shape[0] * num_labels)
        F.v_labels = counts.reshape(v.shape[0], num_labels).argmax(dim=1)
        F.connected_faces, F.edges = rmesh.find_connected_faces(f)
        F.body_v, F.cloth_v = v[F.v_labels == 0], v[F.v_labels == 1]
        geometry_obj._init_basedeform(v, f, F.body_v, F.cloth_v)
        geometry_obj._init_use_body_nonrigid_deform()
        train.FLAGS = F            # train.py's combine_mask (:340) reads the module-global FLAGS that `__main__` defines (:1566)
        keep = a.iters
        train.range = lambda n: (list(builtins.range(keep)) + [n - 1]) if n > keep + 1 else builtins.range(n)     # see the module docstring
        before = [p.detach().clone() for p in geometry_obj.nonrigid.parameters()]
        target = data.collate([data[0]])
        ret = train.optimize_mesh_seq(0, 1, target, None, glctx, geometry_obj, mat, None, data, data, F, pass_idx=0, warmup_iter=0,
                                      log_interval=1, pass_name='pass1', optimize_light=False, save_path=out)
        del train.range
        moved = max(float((p.detach() - b).abs().max()) for p, b in zip(geometry_obj.nonrigid.parameters(), before))
        print('optimize_mesh_seq returned', type(ret).__name__, '; non-rigid network moved by', moved)
        assert moved > 0, 'the optimiser of train.py did not update the non-rigid network'
        dump = np.load(os.path.join(out, 'delta', '1.npz'))
        assert dump['delta'].shape == (v.shape[0], 3) and dump['visible_triangles'].ndim == 1 and dump['visible_triangles'].size > 0
        assert os.path.exists(os.path.join(out, 'fine_all_1.ply')) and os.path.exists(os.path.join(out, 'tmp_all_1.ply'))
        assert finite()
    if ctx is not None:
        ctx.__exit__(None, None, None)
    print('OK')


if __name__ == '__main__':
    main()
