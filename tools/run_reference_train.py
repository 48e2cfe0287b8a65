"""Drive the REFERENCE's own train.py on top of this build -- the drop-in claim of INTEGRATION.md section 2, executed.

    PYTHONPATH=<repo>/d3human-code_amd:<reference root>  python tools/run_reference_train.py [--iters 2] [--emulator] [--res 32] [--grid 6]

What runs is the reference's `train.optimize_mesh_init` (train.py:544-832: its optimiser groups, LambdaLR schedulers, DataLoader loop,
prepare_batch_init, tick_init call, total = reg + normal + msk, backward, encoder-gradient scaling, Adam steps, clamp, stream sync) with
`geometry`, `render`, `deform`, `nvdiffrast`, `tinycudann`, `kaolin`, `pytorch3d`, `ssim_loss`, `lap_loss` resolving to this build and
everything else (`dataset.dataset_split`, `render.{material,texture,light}`, `denoiser`, `script`) to the reference's files.  Inputs the
repository does not ship are synthetic: the dataset object (a seeded stand-in with the reference Dataset_split's `collate` and target
keys), the SMPL-X model (d3h.synth), the tet grid.  Third-party modules that train.py imports at the top but the init stage never calls
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
    assert 'd3human-code_amd' not in where['train'] and 'd3human-code_amd' not in where['dataset.dataset_split'] and \
        'd3human-code_amd' not in where['render.material']

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

    geometry_obj = train.HmSDFTetsGeometry(2 * a.grid, 1.0, F)                      # == geometry.hmsdf of this build
    mat = train.initial_guess_material(geometry_obj, True, F, None)
    mat['no_perturbed_nrm'] = True
    mv, mvp, campos = synth.camera(a.res, dist=3.0)
    H = W = a.res
    yy, xx = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing='ij')
    msk = ((((xx - 0.5 * W) / (0.2 * W)) ** 2 + ((yy - 0.45 * H) / (0.3 * H)) ** 2) < 1).float()[None, ..., None]
    img = torch.cat([torch.tensor([0.55, 0.45, 0.40]).expand(1, H, W, 3) * msk, msk], -1)
    nrm = torch.nn.functional.normalize(torch.stack([(xx - 0.5 * W) / W, -(yy - 0.45 * H) / H, torch.ones_like(xx)], -1), dim=-1)[None] * msk

    class Data(torch.utils.data.Dataset):          # the keys of Dataset_split.__getitem__ (dataset_split.py:255-283)
        def __len__(self):
            return a.iters + 1

        def __getitem__(self, i):
            t = lambda x: torch.from_numpy(x)[None]
            return {'mv': t(mv), 'mvp': t(mvp), 'campos': t(campos), 'resolution': [H, W], 'spp': 1, 'idx': 0,
                    'all_img': img.clone(), 'cloth_img': img.clone(), 'body_img': img.clone(), 'all_normal': nrm.clone(),
                    'cloth_normal': nrm.clone(), 'body_normal': nrm.clone()}
    data = Data()
    data.collate = dataset.dataset_split.Dataset_split.collate.__get__(data)       # the reference's own collate (dataset_split.py:285-311)
    out = a.out or os.path.join('/tmp', 'd3h_train_drive')
    os.makedirs(out, exist_ok=True)
    before = [p.detach().clone() for p in geometry_obj.sdf_net.parameters()]
    ret = train.optimize_mesh_init(None, dr.RasterizeGLContext(), geometry_obj, mat, None, data, data, F, warmup_iter=1, log_interval=1,
                                   pass_idx=0, pass_name='init', optimize_light=False, optimize_geometry=True, visualize=False,
                                   save_path=out)
    moved = max(float((p.detach() - b).abs().max()) for p, b in zip(geometry_obj.sdf_net.parameters(), before))
    print('optimize_mesh_init returned', type(ret).__name__, '; SDF weights moved by', moved)
    assert moved > 0, 'the optimiser of train.py did not update the SDF network'
    assert all(torch.isfinite(p).all() for p in geometry_obj.parameters())
    if ctx is not None:
        ctx.__exit__(None, None, None)
    print('OK')


if __name__ == '__main__':
    main()
