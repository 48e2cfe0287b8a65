"""Synthetic render-and-fit scene + the per-iteration step of the init stage (train.py:651-790), used by bench.py,
__graft_entry__.smoke() and the tests.  Everything is seeded and generated on the device (no datasets offline).

`Scene.step()` is one training iteration exactly as the reference's loop body: zero_grad x3 -> tick_init -> total = reg + normal
+ msk (train.py:718; + the SSIM term of BASELINE config 3 when enabled) -> backward -> encoder grad / 8 (train.py:747-748) ->
three Adam steps with the LambdaLR schedule (train.py:573-620,759-768) -> clamp_deform (train.py:788) -> stream sync (train.py:789).
In the data-parallel mode one flat fp32 bucket of all shared-parameter gradients is all-reduced (RCCL) before the Adam steps.
"""
import os
import types

import numpy as np
import torch

from . import synth


def make_flags(res=512, grid_n=32, n_frames=1, device='cuda', seed=0, prefit_steps=300, iters=2001, body_verts=10475, ssim_weight=0.0,
               visualize_watertight=False, render_buffers=None, sdf_fn=None, frame_seed=1234):
    F = types.SimpleNamespace()
    F.device = device
    F.train_res = [res, res]
    F.texture_res = [res, res]
    F.iter = iters
    F.spp = 1
    F.gender = 'neutral'
    F.boxscale = [1, 1, 1]
    F.skip_in, F.n_freq, F.n_hidden, F.d_hidden, F.use_float16 = [3], 6, 6, 256, False
    F.use_sdf_mlp, F.use_msdf_mlp, F.use_eikonal, F.eikonal_scale = True, False, True, None
    F.sdf_regularizer = 0.2
    F.use_nonrigid_deform = False
    F.use_tanh_deform = False
    F.visualize_watertight = visualize_watertight
    F.n_images = n_frames
    F.out_dir = None
    F.sdf_mlp_pretrain_smpl_steps = prefit_steps
    F.ssim_weight = ssim_weight
    F.render_buffers = render_buffers
    F.use_mesh_msdf_reg, F.msdf_reg_open_scale, F.msdf_reg_close_scale = True, 1e-6, 3e-6      # train.py:1555-1556,1616
    F.lambda_kd, F.lambda_ks, F.lambda_nrm, F.lambda_chroma = 0.1, 0.05, 0.025, 0.0               # train.py:1594-1598
    F.learning_rate = [0.03, 0.005]
    F.kd_min, F.kd_max = [0.0, 0.0, 0.0, 0.0], [1.0, 1.0, 1.0, 1.0]
    F.ks_min, F.ks_max = [0.0, 0.001, 0.0], [0.0, 1.0, 1.0]
    # synthetic stand-ins for the un-shipped inputs
    F.smplx_model_dict = synth.make_body_model(n_verts=body_verts, seed=seed)
    F.tet_grid = synth.kuhn_grid(grid_n)
    F.sdf_init_fn = sdf_fn if sdf_fn is not None else (lambda x: synth.body_sdf(x))
    g = torch.Generator().manual_seed(seed + 100)
    dev = device
    F.shape_param = torch.zeros(1, 100, device=dev)
    F.expr_optim = torch.zeros(n_frames, 50, device=dev)
    F.body_pose_optim = synth.poses(n_frames, seed=frame_seed).to(dev)
    F.root_pose_optim = torch.zeros(n_frames, 3, device=dev)
    F.jaw_pose_optim = torch.zeros(n_frames, 3, device=dev)
    F.trans_optim = torch.zeros(n_frames, 3, device=dev).requires_grad_(True)       # the one pose tensor the init stage optimises (Appendix A)
    F.rhand_pose_optim, F.lhand_pose_optim = torch.zeros(n_frames, 45, device=dev), torch.zeros(n_frames, 45, device=dev)
    F.leye_pose_optim, F.reye_pose_optim = torch.zeros(n_frames, 3, device=dev), torch.zeros(n_frames, 3, device=dev)
    F.face_offset = F.joint_offset = F.locator_offset = None
    return F


from .optim import LambdaLR as _LambdaLR, make_optimizers as _make_optimizers, make_fused_optimizer as _make_fused      # noqa: E402

FUSED_OPTIMIZER = os.environ.get('D3H_FUSED_OPTIMIZER', '1') != '0'      # one-launch Adam (d3h.optim.FusedAdam) instead of two torch.optim.Adam
GRAD_ARENA = os.environ.get('D3H_GRAD_ARENA', '1') != '0'                # frame-parallel: leaf gradients produced inside the all-reduce bucket (d3h/gradarena.py)


class _ZeroOffset(torch.nn.Module):
    def forward(self, x, code):
        return torch.zeros_like(x)


class Scene:
    def __init__(self, res=512, grid_n=32, n_frames=1, device='cuda', seed=0, prefit_steps=300, loss_set='full', body_verts=10475,
                 visualize_watertight=False, dist_world=1, dist_rank=0, sdf_fn=None, flags_hook=None, frame_seed=1234, lpips=None,
                 split_partition=True, sdf_state=None):
        import nvdiffrast.torch as dr
        from geometry.hmsdf import HmSDFTetsGeometry
        from render.mlptexture import MLPTexture3D
        torch.manual_seed(seed)
        self.loss_set = loss_set
        want = {'mask': ('shaded',)}.get(loss_set)          # config 2 reads the mask only; otherwise tick_*'s default: the buffers it reads
        self.FLAGS = make_flags(res, grid_n, n_frames, device, seed, prefit_steps, ssim_weight=(1.0 if loss_set == 'full' else 0.0),
                                visualize_watertight=visualize_watertight, render_buffers=want, body_verts=body_verts, sdf_fn=sdf_fn, frame_seed=frame_seed)
        F = self.FLAGS
        if loss_set == 'seq':
            F.use_nonrigid_deform = True                                       # train.py:1617
            F.sdf_deform_pretrain_steps = 300 if prefit_steps == 0 else prefit_steps   # zero-offset pre-fit (hmsdf.py:293-308)
            F.deform_checkpoint = None
            F.sdf_mlp_pretrain_smpl_steps = 0                                  # the SDF network is not evaluated in this stage
        if lpips is not None:                     # split stage with the LPIPS term (BASELINE config 5): an lpips.LPIPS module
            F.lpips_fn, F.lpips_weight = lpips.to(device), 1.0
            for p_ in F.lpips_fn.parameters():          # a fixed metric: neither the trunk nor the calibrated linear layers are trained
                p_.requires_grad_(False)
        if flags_hook is not None:
            flags_hook(F)
        if sdf_state is not None:
            F.sdf_mlp_pretrain_smpl_steps = 0      # the SDF network comes from `sdf_state`: no pre-fit (whose backward sums with float atomics)
        self.device = torch.device(device)
        self.glctx = dr.RasterizeGLContext()
        self.geometry = HmSDFTetsGeometry(2 * grid_n, 1.0, F)          # grid_res only scales max_displacement (Appendix A)
        if sdf_state is not None:
            self._load_sdf_state(sdf_state)
        t = lambda v: torch.tensor(v, dtype=torch.float32, device=device)
        mlp_min = torch.cat((t(F.kd_min)[0:3], t(F.ks_min)))
        mlp_max = torch.cat((t(F.kd_max)[0:3], t(F.ks_max)))
        self.material = {'kd_ks': MLPTexture3D(self.geometry.getAABB(), channels=6, min_max=[mlp_min, mlp_max]).to(device), 'bsdf': 'pbr'}
        mv, mvp, campos = synth.camera(res)
        self.mvp = torch.from_numpy(mvp).to(device)[None].expand(n_frames, -1, -1).contiguous()
        self.mv = torch.from_numpy(mv).to(device)[None].expand(n_frames, -1, -1).contiguous()
        self.campos = torch.from_numpy(campos).to(device)[None].expand(n_frames, -1).contiguous()
        self.n_frames, self.res = n_frames, res
        self.world, self.rank = dist_world, dist_rank
        if loss_set == 'seq':
            self._setup_seq()
        else:
            self._make_targets()
            if loss_set == 'split' and split_partition:
                self._make_split_partition()
            self._make_optimizers()
        self.it = 0

    # ---- a state no GPU arithmetic has produced (the whole-tick parity tests: every box compares the SAME scene) -----------------------
    @torch.no_grad()
    def _load_sdf_state(self, sdf_state):
        """`sdf_state`: path of an .npz / a dict {net.<i>.weight|bias: array} with the SDF network's state_dict keys (geometry/mlp.py:13-31),
        e.g. tests/golden/parity_state_sdf.npz (fitted on the CPU by tools/gen_parity_state.py).  The initial mSDF is re-drawn from a CPU
        generator (hmsdf.py:311's formula) so that it does not depend on the device's random stream either."""
        g = self.geometry
        if isinstance(sdf_state, (str, os.PathLike)):
            sdf_state = dict(np.load(sdf_state))
        sd = {k: torch.as_tensor(np.asarray(v), dtype=torch.float32, device=self.device) for k, v in sdf_state.items() if k.startswith('net.')}
        g.sdf_net.load_state_dict(sd)
        gen = torch.Generator().manual_seed(4242)
        g.msdf.data.copy_((torch.rand(g.verts.shape[0], generator=gen) - 0.01).clamp(-1, 1).to(self.device))

    @torch.no_grad()
    def perturb_state_seeded(self, seed=0, deform_amp=0.35, trans_amp=0.004):
        """Moves the scene off its initial point WITHOUT optimiser steps (their gradients are summed by float atomics: not reproducible):
        `deform` = a smooth seeded field (three sine waves per component, amplitude `deform_amp` of the clamp range [-1, 1]), `trans` = a
        seeded offset.  Evaluated in float64 on the host from the grid coordinates: the same bits on every box."""
        g, F = self.geometry, self.FLAGS
        rng = np.random.default_rng(9000 + seed)
        x = g.verts.detach().cpu().double().numpy()
        d = np.zeros_like(x)
        for c in range(3):
            for _ in range(3):
                k = rng.uniform(2.0, 9.0, size=3) * rng.choice([-1.0, 1.0], size=3)
                d[:, c] += np.sin(x @ k + rng.uniform(0, 2 * np.pi)) / 3.0
        g.deform.data.copy_(torch.from_numpy((deform_amp * d).astype(np.float32)).to(self.device))
        F.trans_optim.data.copy_(torch.from_numpy(rng.uniform(-trans_amp, trans_amp, size=tuple(F.trans_optim.shape)).astype(np.float32)).to(self.device))

    @torch.no_grad()
    def set_kinkfree_texture(self, seed=0):
        """a texture state WITHOUT ReLU kinks: positive table entries and positive first / second layer weights keep every hidden
        pre-activation of the texture MLP (mlptexture.py:18-41: no biases) strictly positive, so any two implementations evaluate the
        piecewise-linear network inside one linear piece whatever their summation order.  Drawn from a CPU generator."""
        tex = self.material['kd_ks']
        gen = torch.Generator().manual_seed(7000 + seed)
        tab = tex.encoder.params
        tab.data.copy_((torch.rand(tab.shape, generator=gen) * 0.30 + 0.05).to(tab.device))
        for i in (0, 2):
            w = tex.net.net[i].weight
            k = 1.0 / w.shape[1] ** 0.5
            w.data.copy_((torch.rand(w.shape, generator=gen) * k + 0.02).to(w.device))
        w = tex.net.net[4].weight
        k = 1.0 / w.shape[1] ** 0.5
        w.data.copy_(((torch.rand(w.shape, generator=gen) * 2 - 1) * k).to(w.device))

    # ---- seq stage: a fixed-topology body + garment mesh driven by the non-rigid network (train.py:1865-1926, 1246-1460) --------------
    @torch.no_grad()
    def _setup_seq(self):
        """Synthetic stand-in for `merge_body_cloth.obj` + Dataset_split labels: body = the zero level set of the analytic SDF, garment
        = its 0.03 offset shell over the torso band (an open surface), both extracted by the marching-tets kernels; then the label /
        connectivity preparation of train.py:1885-1911 and targets rendered from the same mesh at a displaced translation."""
        from render import mesh as rmesh
        from geometry.hmsdf import _flag
        F, dev, g = self.FLAGS, self.device, self.geometry
        sdf = F.sdf_init_fn(g.verts).reshape(-1).to(dev)
        ones = torch.ones_like(sdf)
        _, _, _, _, _, ex_b = g.gshell_tets(g.verts, sdf, ones, g.indices)
        band = ((g.verts[:, 1] > -0.75) & (g.verts[:, 1] < 0.05)).float() * 2 - 1       # garment exists where msdf > 0
        cv, cf, _, _, _, ex_c = g.gshell_tets(g.verts, sdf - 0.03, band, g.indices)
        bv, bf = ex_b['vertices_watertight'], ex_b['faces_watertight']
        used = torch.unique(cf)                                                       # drop unreferenced rows of verts_aug
        remap = torch.full((cv.shape[0],), -1, dtype=torch.long, device=dev)
        remap[used] = torch.arange(used.shape[0], device=dev)
        cv, cf = cv[used], remap[cf]
        v = torch.cat([bv, cv]).contiguous()
        f = torch.cat([bf, cf + bv.shape[0]]).long().contiguous()
        face_labels = torch.cat([torch.zeros(bf.shape[0], dtype=torch.long, device=dev), torch.ones(cf.shape[0], dtype=torch.long, device=dev)])
        F.v, F.f, F.face_labels = v, f, face_labels
        F.body_f, F.cloth_f = f[face_labels == 0], f[face_labels == 1]
        # train.py:1889-1898: a vertex takes the label most of its incident face corners carry
        nl = int(face_labels.max().item()) + 1
        counts = torch.bincount(f.reshape(-1) * nl + face_labels[:, None].expand(-1, 3).reshape(-1), minlength=v.shape[0] * nl)
        F.v_labels = counts.reshape(v.shape[0], nl).argmax(dim=1)
        F.connected_faces, F.edges = rmesh.find_connected_faces(f)
        F.body_v, F.cloth_v = v[F.v_labels == 0], v[F.v_labels == 1]
        # collision_loss gathers body_pos[body_faces] with body_pos = the label-0 vertices (hmsdf.py:799-805): re-index the body faces
        to_body = torch.full((v.shape[0],), -1, dtype=torch.long, device=dev)
        to_body[F.v_labels == 0] = torch.arange(int((F.v_labels == 0).sum()), device=dev)
        F.body_f = to_body[F.body_f]
        g._init_basedeform(v, f, F.body_v, F.cloth_v)
        # targets: the undeformed merged mesh at a displaced translation; cloth / body masks from its face labels
        tr = F.trans_optim.detach().clone()
        F.trans_optim = tr + torch.tensor([0.02, 0.01, 0.0], device=dev)
        save = g.nonrigid
        g.nonrigid = _ZeroOffset()
        tgt = self.target(torch.zeros(1, self.res, self.res, 3, device=dev))
        d = g.render_seq(self.glctx, tgt, None, self.material, buffers=('shaded', 'geometric_normal'))
        g.nonrigid = save
        F.trans_optim = tr
        b = d['all_mesh_buffers']
        albedo = torch.tensor([0.55, 0.45, 0.40], device=dev)

        def img(mask):
            m = (mask[..., None] > 0.5).float()
            return torch.cat([albedo.expand(*m.shape[:-1], 3) * m, m], -1).contiguous()
        self.all_img, self.cloth_img, self.body_img = img(d['all_mask']), img(d['cloth_mask']), img(d['body_mask'])
        n = b['geometric_normal'][..., :3] * torch.tensor([1.0, -1.0, -1.0], device=dev)
        self.all_normal = (torch.nn.functional.normalize(n, dim=-1) * self.all_img[..., 3:]).contiguous()
        # optimisers (train.py:1295-1312): non-rigid network + cond codes at lr_pos * 1e-2, material at lr_mat; warm-up 0 (train.py:1926)
        self._build_optimizers('seq', warmup_iter=0)

    def step_seq(self):
        """one iteration of the seq stage (train.py:1364-1460): tick_seq, total = 250 normal + 0.1 reg + masks + 1e6 laplacian +
        1e5 collision + 1e3 normal-consistency + delta (train.py:1412-1421)"""
        bg = torch.rand(1, self.res, self.res, 3, device=self.device)
        tgt = self.target(bg)
        tgt.update({'cloth_img': self.cloth_img, 'body_img': self.body_img})
        self._zero_grad()
        r = self.geometry.tick_seq(self.glctx, tgt, None, self.material, self.loss_fn, self.it, None, t='all')
        total = 250 * r['normal_loss'] + 0.1 * r['reg_loss'] + (r['body_msk_loss'] + r['cloth_msk_loss'] + r['all_msk_loss']) + \
            1000000 * r['laplacian_loss'] + 100000 * r['colli_loss'] + 1000 * r['nds_normal_loss'] + r['delta_loss']
        total.backward()
        self._optimizer_step(clamp=False)
        self.it += 1
        self.last = {k: v.detach() for k, v in r.items() if torch.is_tensor(v) and v.dim() == 0}
        self.last['total'] = total.detach()
        return self.last

    # ---- targets: the pre-fit body rendered once at a displaced pose (analytic-humanoid stand-in for the dataset) -----------------
    @torch.no_grad()
    def _make_targets(self):
        F, dev = self.FLAGS, self.device
        tr = F.trans_optim.detach().clone()
        F.trans_optim = (tr + torch.tensor([0.02, 0.01, 0.0], device=dev)).requires_grad_(False)
        tgt = self.target(torch.zeros(self.n_frames, self.res, self.res, 3, device=dev))
        save_want = F.render_buffers
        F.render_buffers = ('shaded', 'geometric_normal')
        d = self.geometry.render_init(self.glctx, tgt, None, self.material)
        F.render_buffers = save_want
        F.trans_optim = tr.requires_grad_(True)
        b = d['buffers']
        mask = (b['shaded'][..., 3:] > 0.5).float()
        albedo = torch.tensor([0.55, 0.45, 0.40], device=dev)
        self.all_img = torch.cat([albedo.expand_as(b['shaded'][..., :3]) * mask, mask], -1).contiguous()
        n = b['geometric_normal'][..., :3] * torch.tensor([1.0, -1.0, -1.0], device=dev)
        self.all_normal = (torch.nn.functional.normalize(n, dim=-1) * mask).contiguous()

    @torch.no_grad()
    def _make_split_partition(self):
        """The split stage as it looks once the mSDF has learnt the garment / body partition (the reference starts it from the init
        stage's mSDF, positive almost everywhere -- hmsdf.py:311 -- where the garment pass extracts the whole surface and the body pass,
        which negates the mSDF, a sliver: that is the first iterations, not the stage): mSDF = +0.5 on the grid vertices of the torso
        band (the garment: hmSDF_Tets keeps the surface where the mSDF is positive), -0.5 elsewhere (the body pass keeps that), with a
        seeded +-0.02 ripple so that the cut polygons are not axis-aligned; garment and body targets rendered from the two extractions
        at the displaced pose of _make_targets (dataset/dataset_split.py:255-283: cloth_img / body_img / *_normal)."""
        F, dev, g = self.FLAGS, self.device, self.geometry
        y = g.verts[:, 1]
        band = ((y > -0.75) & (y < 0.05)).float() * 2 - 1
        gen = torch.Generator().manual_seed(17)
        g.msdf.data.copy_((0.5 * band + 0.04 * (torch.rand(y.shape[0], generator=gen).to(dev) - 0.5)).clamp(-1, 1))
        tr = F.trans_optim.detach().clone()
        F.trans_optim = (tr + torch.tensor([0.02, 0.01, 0.0], device=dev)).requires_grad_(False)
        tgt = self.target(torch.zeros(self.n_frames, self.res, self.res, 3, device=dev))
        albedo = {'cloth': torch.tensor([0.30, 0.50, 0.65], device=dev), 'body': torch.tensor([0.55, 0.45, 0.40], device=dev)}
        self.split_faces = {}
        for typ in ('cloth', 'body'):
            g._sweep_cache = None
            d = g.render_split(self.glctx, tgt, None, self.material, typ, buffers=('shaded', 'geometric_normal'))
            b = d['buffers']
            mask = (b['shaded'][..., 3:] > 0.5).float()
            img = torch.cat([albedo[typ].expand_as(b['shaded'][..., :3]) * mask, mask], -1).contiguous()
            n = b['geometric_normal'][..., :3] * torch.tensor([1.0, -1.0, -1.0], device=dev)
            setattr(self, typ + '_img', img)
            setattr(self, typ + '_normal', (torch.nn.functional.normalize(n, dim=-1) * mask).contiguous())
            self.split_faces[typ] = int(d['imesh'].t_pos_idx.shape[0])
        F.trans_optim = tr.requires_grad_(True)

    def target(self, background):
        ai, an = getattr(self, 'all_img', None), getattr(self, 'all_normal', None)
        # without a garment / body partition (init stage; split_partition=False) the scene has one surface and the garment and body targets
        # coincide with it (dataset/dataset_split.py:255-283 keys)
        return {'idx': list(range(self.n_frames)), 'mv': self.mv, 'mvp': self.mvp, 'campos': self.campos,
                'resolution': [self.res, self.res], 'spp': 1, 'background': background,
                'all_img': ai, 'all_normal': an,
                'cloth_img': getattr(self, 'cloth_img', ai), 'cloth_normal': getattr(self, 'cloth_normal', an),
                'body_img': getattr(self, 'body_img', ai), 'body_normal': getattr(self, 'body_normal', an)}

    # ---- optimisers (train.py:573-620) ---------------------------------------------------------------------------------------------
    def _make_optimizers(self):
        """the reference's groups / learning rates / schedule for this stage (d3h/optim.py <- train.py:569-620, 862-912)"""
        self._build_optimizers('split' if self.loss_set == 'split' else 'init', warmup_iter=300)

    def _build_optimizers(self, stage, warmup_iter):
        F, mat = self.FLAGS, self.material['kd_ks']
        if FUSED_OPTIMIZER:
            # both optimisers of the stage, the encoder-gradient scale (train.py:747-748) and clamp_deform (train.py:788) in one launch
            self.opt, sched = _make_fused(stage, self.geometry, mat, F, warmup_iter=warmup_iter)
            self.opt_geo, self.opt_mat, self.sched = self.opt, None, [sched]
            groups = self.opt.param_groups
        else:
            self.opt = None
            self.opt_geo, self.opt_mat, self.sched = _make_optimizers(stage, self.geometry, mat.parameters(), F, warmup_iter=warmup_iter,
                                                                      fused=self.device.type == 'cuda')
            groups = self.opt_geo.param_groups + self.opt_mat.param_groups
        # data-parallel bucket: every parameter the stage's optimisers update except the per-frame pose rows (owned by the frame's rank)
        pose = {id(F.trans_optim)}
        self.shared_params = [p for grp in groups for p in grp['params'] if id(p) not in pose]

    def _zero_grad(self):
        self.opt_geo.zero_grad(set_to_none=True)
        if self.opt_mat is not None:
            self.opt_mat.zero_grad(set_to_none=True)
        # parameters that receive gradients but belong to no optimiser group of this stage (the SDF network in the split stage,
        # train.py:896-902): the reference lets their .grad accumulate unread; dropping it here keeps AccumulateGrad on its no-copy path
        gp = self.__dict__.get('_geo_params')
        if gp is None:          # (the module tree is fixed after construction; walking it costs ~80 us per step)
            gp = self._geo_params = list(self.geometry.parameters())
        for p in gp:
            p.grad = None
        if self.world > 1 and GRAD_ARENA:
            self._grad_arena().begin()         # frame-parallel: this step's shared-parameter gradients are produced inside the all-reduce bucket

    def _optimizer_step(self, clamp=True):
        """train.py:747-788: encoder gradient / 8, (data-parallel: the gradient bucket), the Adam steps + schedulers, clamp_deform"""
        if self.opt is not None:
            if self.world > 1:
                self.allreduce_grads()
            self.opt.step(); self.sched[0].step()
            return
        enc = self.material['kd_ks'].encoder.params
        if enc.grad is not None:
            enc.grad /= 8.0
        if self.world > 1:
            self.allreduce_grads()
        self.opt_geo.step(); self.sched[0].step()
        self.opt_mat.step(); self.sched[1].step()
        if clamp:
            with torch.no_grad():
                self.geometry.clamp_deform()

    def loss_fn(self, img, ref):
        from render import renderutils as ru
        return ru.image_loss(img, ref, loss='l1', tonemapper='log_srgb')           # train.py:81 'logl1'

    if os.environ.get('D3H_SCENE_PLAIN_LOSS') != '1':          # '1': a bare callable, as train.py:75-87 builds it (tick_* probe it, renderutils.loss_spec)
        loss_fn.d3h_spec = ('l1', 'log_srgb')          # lets tick_* evaluate it inside the fused per-pixel loss pass

    def step(self):
        """one init-stage iteration.  D3H_MAIN_PRIORITY=1 (experiment): the whole step runs on a high-priority HIP stream, so that its
        HBM-bound render / loss kernels win the workgroup slots against the eikonal chain of the (normal-priority) side stream"""
        if os.environ.get('D3H_MAIN_PRIORITY') == '1' and self.device.type == 'cuda':
            if getattr(self, '_hp', None) is None:
                self._hp = torch.cuda.Stream(device=self.device, priority=-1)
                self._hp.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self._hp):
                return self._step()
        return self._step()

    def _step(self):
        F = self.FLAGS
        it = self.it
        bg = torch.rand(self.n_frames, self.res, self.res, 3, device=self.device)   # random background per iteration (train.py:653)
        tgt = self.target(bg)
        self._zero_grad()
        r = self.geometry.tick_init(self.glctx, tgt, None, self.material, self.loss_fn, it, None)
        if self.loss_set == 'mask':
            total = r['msk_loss']
        elif 'd3h_total' in r:
            total = r['d3h_total']                     # the same sum, formed inside tick_init's affine loss head
        else:
            total = r['reg_loss'] + r['normal_loss'] + r['msk_loss'] + r.get('ssim_loss', 0.0)
        total.backward()
        self._optimizer_step()
        self.it += 1
        self.last = {k: v.detach() for k, v in r.items()}
        self.last['total'] = total.detach()
        return self.last

    def step_split(self):
        """one iteration of the split stage (train.py:1035-1100): tick_split for the garment and for the body (hmSDF_Tets with the mSDF
        negated), total = sum over both of img + normal + reg + 10 * msk (train.py:1050,1067,1087)"""
        it = self.it
        bg = torch.rand(self.n_frames, self.res, self.res, 3, device=self.device)
        tgt = self.target(bg)
        self._zero_grad()
        total = 0.0
        last = {}
        self.FLAGS.share_sdf_sweep = getattr(self, 'share_sweep', True)   # one SDF sweep (forward + backward) for both extractions
        self.geometry._sweep_cache = None
        for typ in ('cloth', 'body'):
            r = self.geometry.tick_split(self.glctx, tgt, None, self.material, self.loss_fn, it, None, type=typ)
            total = total + r['img_loss'] + r['normal_loss'] + r['reg_loss'] + 10 * r['msk_loss']
            last.update({f'{typ}_{k}': v.detach() for k, v in r.items()})
        total.backward()
        self._optimizer_step()
        self.it += 1
        last['total'] = total.detach()
        self.last = last
        return last

    def enable_work_sharding(self, eikonal_total=None):
        """Frame-parallel runs: from now on the frame-INDEPENDENT work of the step is split over the ranks instead of replicated
        (d3h/dist_ops.py): each rank evaluates 1/W of the SDF sweep over the tet grid (values all-gathered, their gradient
        reduce-scattered) and draws ceil(S / W) of the S surface samples of the eikonal term (hmsdf.py:714: S = 50 000; the mean over
        the ranks of the per-rank means is the same estimator), from its own random stream.  Call it once the shared parameters are
        identical on every rank (after the broadcast): the shards are only consistent then."""
        from . import dist_ops as D
        if self.world > 1 and os.environ.get('D3H_SHARD_SWEEP', '1') != '0':
            self.FLAGS.sdf_shard = (self.rank, self.world)
        if self.world > 1 and os.environ.get('D3H_SHARD_EIKONAL', '1') != '0':
            total = int(eikonal_total if eikonal_total is not None else getattr(self.FLAGS, 'eikonal_samples', 50000))
            self.FLAGS.eikonal_samples = D.samples_per_rank(total, self.world)
            torch.manual_seed(0x5eed + 1000003 * (self.rank + 1))         # every rank its own samples / backgrounds (all generators)

    def disable_work_sharding(self, eikonal_total=None):
        """back to replicated frame-independent work (every rank the whole sweep and all S samples; one collective per step).  The ranks'
        random streams stay distinct (different backgrounds per rank are fine: the gradients are averaged)."""
        self.FLAGS.sdf_shard = None
        self.FLAGS.eikonal_samples = int(eikonal_total if eikonal_total is not None else 50000)

    def enable_sweep_sharding(self):
        """the sweep half of enable_work_sharding alone (the round-1..3 option)"""
        if self.world > 1 and os.environ.get('D3H_SHARD_SWEEP', '1') != '0':
            self.FLAGS.sdf_shard = (self.rank, self.world)

    # ---- virtual rank: ONE process stands in for rank `dist_rank` of a `dist_world`-rank job (bench.py --as-rank-of) --------------------
    @torch.no_grad()
    def refresh_virtual(self):
        """what the other ranks would contribute to the sdf all-gather: a full sweep of the current parameters (outside any timed region)"""
        from . import dist_ops as D
        g = self.geometry
        if getattr(self.FLAGS, 'sdf_shard', None) is not None and D.virtual() is not None:
            D.set_virtual_full(g.sdf_net(g.verts, deform=g.deform, disp=g.max_displacement))

    def freeze_learning(self):
        """learning rates to zero from now on: every kernel of the step still runs (Adam included), the parameters stay where they are --
        the virtual-rank mode times a step whose foreign sweep shards stay exact"""
        for sc in self.sched:
            if hasattr(sc, 'base'):                      # d3h.optim.LambdaLR
                sc.base = [0.0 for _ in sc.base]
            if hasattr(sc, 'base_lrs'):                  # torch.optim.lr_scheduler.LambdaLR (D3H_FUSED_OPTIMIZER=0)
                sc.base_lrs = [0.0 for _ in sc.base_lrs]
            if hasattr(sc, '_apply'):
                sc._apply()
        for grp in (self.opt_geo.param_groups + (self.opt_mat.param_groups if self.opt_mat is not None else [])):
            grp['lr'] = 0.0

    # ---- frame-parallel data parallelism: ONE flat fp32 bucket, one all-reduce (RCCL over xGMI), mean over the ranks ---------------------
    def _bucket_members(self):
        """EVERY shared parameter that requires a gradient, whether or not this rank's backward produces one (torch DDP's rule); the 16
        tensors of the fused SDF network adjacent and in the order its gradient kernels write them (d3h.sdf_mlp arena order), so that
        their summed gradient lands in the bucket with one copy"""
        ps = [p for p in self.shared_params if p.requires_grad]          # per-frame pose rows (trans_optim) belong to the frame's rank
        net = getattr(getattr(self, 'geometry', None), 'sdf_net', None)
        if net is not None and getattr(net, 'fused', False):
            from .sdf_mlp import _ARENA_PERM
            prm = net._params()
            block = [prm[k] for k in _ARENA_PERM]
            ids = {id(p) for p in block}
            if all(any(q is p for q in ps) for p in block):
                first = min(i for i, q in enumerate(ps) if id(q) in ids)
                rest = [q for q in ps if id(q) not in ids]
                ps = rest[:first] + block + rest[first:]
        return ps

    def _grad_arena(self):
        from .gradarena import GradArena
        a = getattr(self, '_arena', None)
        if a is None:
            a = self._arena = GradArena(self._bucket_members())
            self._bucket = a.params
            self.bucket_bytes = 4 * a.numel
            self._bucket_checked = False
        return a

    def allreduce_grads(self):
        """ONE flat fp32 bucket of every shared-parameter gradient -> all_reduce(mean) in place.  The bucket is the step's gradient arena
        (d3h/gradarena.py): the kernels that produce the big leaf gradients (deform, grid table, texture weights, the flat SDF vector) wrote
        into it during the backward, so there is no flatten copy before the collective and none after it -- `.grad` of every member IS its
        slice of the bucket.  A member without a local gradient contributes zeros (an empty garment on one rank, a loss term that
        switches on later), so the ranks can never disagree on the layout or silently keep a local-only gradient; they check once that
        they hold the same layout.  `self.coll_timing` (a list, set by bench.py) collects (start, end) events around the collective on the
        launch stream.  D3H_GRAD_ARENA=0: the round-3 path (cat -> all_reduce -> scale -> copy back)."""
        import torch.distributed as dist
        from . import dist_ops as D
        if not GRAD_ARENA:
            return self._allreduce_grads_flatten()
        a = self._grad_arena()
        if not self._bucket_checked and D.virtual() is None:
            sig = torch.tensor([len(a.params), a.numel], dtype=torch.int64, device=a.flat.device)
            lo, hi = sig.clone(), sig.clone()
            dist.all_reduce(lo, op=dist.ReduceOp.MIN)
            dist.all_reduce(hi, op=dist.ReduceOp.MAX)
            if not (torch.equal(lo, sig) and torch.equal(hi, sig)):
                raise RuntimeError(f'allreduce_grads: the ranks disagree on the gradient bucket ({sig.tolist()} here, min {lo.tolist()}, max {hi.tolist()})')
        self._bucket_checked = True
        flat = a.collect()
        ev = None
        if getattr(self, 'coll_timing', None) is not None and flat.is_cuda:
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
        D.all_reduce_mean(flat, self.world)
        if ev is not None:
            ev[1].record()
            self.coll_timing.append(ev)

    def _allreduce_grads_flatten(self):
        import torch.distributed as dist
        from . import dist_ops as D
        ps = getattr(self, '_bucket', None)
        if ps is None:
            ps = self._bucket_members()
            if D.virtual() is None:
                sig = torch.tensor([len(ps), sum(p.numel() for p in ps)], dtype=torch.int64, device=ps[0].device if ps else 'cpu')
                lo, hi = sig.clone(), sig.clone()
                dist.all_reduce(lo, op=dist.ReduceOp.MIN)
                dist.all_reduce(hi, op=dist.ReduceOp.MAX)
                if not (torch.equal(lo, sig) and torch.equal(hi, sig)):
                    raise RuntimeError(f'allreduce_grads: the ranks disagree on the gradient bucket ({sig.tolist()} here, min {lo.tolist()}, max {hi.tolist()})')
            self._bucket = ps
            self.bucket_bytes = 4 * sum(p.numel() for p in ps)
        for p in ps:
            if p.grad is None:
                p.grad = torch.zeros_like(p)
        grads = [p.grad for p in ps]
        flat = torch.cat([g.reshape(-1) for g in grads])                      # one batched copy kernel
        ev = None
        if getattr(self, 'coll_timing', None) is not None and flat.is_cuda:
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
        D.all_reduce_mean(flat, self.world)
        if ev is not None:
            ev[1].record()
            self.coll_timing.append(ev)
        outs, o = [], 0
        for g in grads:
            n = g.numel()
            outs.append(flat[o:o + n].view_as(g))
            o += n
        torch._foreach_copy_(grads, outs)                                     # one multi-tensor kernel instead of one copy per tensor
