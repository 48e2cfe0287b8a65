"""Synthetic render-and-fit scene + the per-iteration step of the init stage (train.py:651-790), used by bench.py,
__graft_entry__.smoke() and the tests.  Everything is seeded and generated on the device (no datasets offline).

`Scene.step()` is one training iteration exactly as the reference's loop body: zero_grad x3 -> tick_init -> total = reg + normal
+ msk (train.py:718; + the SSIM term of BASELINE config 3 when enabled) -> backward -> encoder grad / 8 (train.py:747-748) ->
three Adam steps with the LambdaLR schedule (train.py:573-620,759-768) -> clamp_deform (train.py:788) -> stream sync (train.py:789).
In the data-parallel mode one flat fp32 bucket of all shared-parameter gradients is all-reduced (RCCL) before the Adam steps.
"""
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
    F.render_buffers_split = ('shaded', 'geometric_normal', 'msdf_image', 'kd', 'kd_grad', 'ks_grad', 'normal_grad')   # what tick_split reads
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
    F.face_offset = F.joint_offset = F.locator_offset = None
    return F


class Scene:
    def __init__(self, res=512, grid_n=32, n_frames=1, device='cuda', seed=0, prefit_steps=300, loss_set='full', body_verts=10475,
                 visualize_watertight=False, dist_world=1, dist_rank=0, sdf_fn=None, flags_hook=None, frame_seed=1234):
        import nvdiffrast.torch as dr
        from geometry.hmsdf import HmSDFTetsGeometry
        from render.mlptexture import MLPTexture3D
        torch.manual_seed(seed)
        self.loss_set = loss_set
        want = {'mask': ('shaded',), 'full': ('shaded', 'geometric_normal', 'msdf_image'),
                'split': ('shaded', 'geometric_normal', 'msdf_image')}.get(loss_set)
        self.FLAGS = make_flags(res, grid_n, n_frames, device, seed, prefit_steps, ssim_weight=(1.0 if loss_set == 'full' else 0.0),
                                visualize_watertight=visualize_watertight, render_buffers=want, body_verts=body_verts, sdf_fn=sdf_fn, frame_seed=frame_seed)
        F = self.FLAGS
        if flags_hook is not None:
            flags_hook(F)
        self.device = torch.device(device)
        self.glctx = dr.RasterizeGLContext()
        self.geometry = HmSDFTetsGeometry(2 * grid_n, 1.0, F)          # grid_res only scales max_displacement (Appendix A)
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
        self._make_targets()
        self._make_optimizers()
        self.it = 0

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

    def target(self, background):
        ai, an = getattr(self, 'all_img', None), getattr(self, 'all_normal', None)
        return {'idx': list(range(self.n_frames)), 'mv': self.mv, 'mvp': self.mvp, 'campos': self.campos,
                'resolution': [self.res, self.res], 'spp': 1, 'background': background,
                'all_img': ai, 'all_normal': an,
                # the synthetic scene has one surface: garment and body targets coincide (dataset/dataset_split.py:255-283 keys)
                'cloth_img': ai, 'cloth_normal': an, 'body_img': ai, 'body_normal': an}

    # ---- optimisers (train.py:573-620) ---------------------------------------------------------------------------------------------
    def _make_optimizers(self):
        F = self.FLAGS
        lr_pos, lr_mat = F.learning_rate

        def lr_schedule(it, fraction=0.02):
            warmup = 300
            return it / warmup if it < warmup else max(0.0, 10 ** (-(it - warmup) * 0.0002))
        deform_p = [p for n, p in self.geometry.named_parameters() if 'deform' in n]
        sdf_p = [p for n, p in self.geometry.named_parameters() if 'sdf' in n]
        other_p = [p for n, p in self.geometry.named_parameters() if 'deform' not in n and 'sdf' not in n]
        groups = [{'params': deform_p, 'lr': lr_pos}, {'params': sdf_p, 'lr': lr_pos * 1e-2}, {'params': other_p, 'lr': lr_pos * 1e-3},
                  {'params': [F.trans_optim], 'lr': lr_pos * 1e-3}]
        fused = self.device.type == 'cuda'
        self.opt_geo = torch.optim.Adam(groups, eps=1e-8, fused=fused)
        self.opt_mat = torch.optim.Adam(self.material['kd_ks'].parameters(), lr=lr_mat, fused=fused)
        self.sched = [torch.optim.lr_scheduler.LambdaLR(o, lr_lambda=lr_schedule) for o in (self.opt_geo, self.opt_mat)]
        self.shared_params = [p for g in groups[:3] for p in g['params']] + list(self.material['kd_ks'].parameters())

    def loss_fn(self, img, ref):
        from render import renderutils as ru
        return ru.image_loss(img, ref, loss='l1', tonemapper='log_srgb')           # train.py:81 'logl1'

    loss_fn.d3h_spec = ('l1', 'log_srgb')          # lets tick_* evaluate it inside the fused per-pixel loss pass

    def step(self):
        F = self.FLAGS
        it = self.it
        bg = torch.rand(self.n_frames, self.res, self.res, 3, device=self.device)   # random background per iteration (train.py:653)
        tgt = self.target(bg)
        self.opt_geo.zero_grad(set_to_none=True)
        self.opt_mat.zero_grad(set_to_none=True)
        r = self.geometry.tick_init(self.glctx, tgt, None, self.material, self.loss_fn, it, None)
        if self.loss_set == 'mask':
            total = r['msk_loss']
        else:
            total = r['reg_loss'] + r['normal_loss'] + r['msk_loss'] + r.get('ssim_loss', 0.0)
        total.backward()
        enc = self.material['kd_ks'].encoder.params
        if enc.grad is not None:
            enc.grad /= 8.0                                                          # train.py:747-748
        if self.world > 1:
            self.allreduce_grads()
        self.opt_geo.step(); self.sched[0].step()
        self.opt_mat.step(); self.sched[1].step()
        with torch.no_grad():
            self.geometry.clamp_deform()
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
        self.opt_geo.zero_grad(set_to_none=True)
        self.opt_mat.zero_grad(set_to_none=True)
        total = 0.0
        last = {}
        for typ in ('cloth', 'body'):
            r = self.geometry.tick_split(self.glctx, tgt, None, self.material, self.loss_fn, it, None, type=typ)
            total = total + r['img_loss'] + r['normal_loss'] + r['reg_loss'] + 10 * r['msk_loss']
            last.update({f'{typ}_{k}': v.detach() for k, v in r.items()})
        total.backward()
        enc = self.material['kd_ks'].encoder.params
        if enc.grad is not None:
            enc.grad /= 8.0
        if self.world > 1:
            self.allreduce_grads()
        self.opt_geo.step(); self.sched[0].step()
        self.opt_mat.step(); self.sched[1].step()
        with torch.no_grad():
            self.geometry.clamp_deform()
        self.it += 1
        last['total'] = total.detach()
        self.last = last
        return last

    # ---- frame-parallel data parallelism: ONE flat fp32 bucket, one all-reduce (RCCL over xGMI), scale by 1/W ------------------------
    def allreduce_grads(self):
        import torch.distributed as dist
        ps = self.shared_params           # per-frame pose rows (trans_optim) belong to the rank that owns the frame
        for p in ps:
            if p.grad is None:
                p.grad = torch.zeros_like(p)
        flat = torch.cat([p.grad.reshape(-1) for p in ps])
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        flat /= self.world
        o = 0
        for p in ps:
            n = p.numel()
            p.grad.copy_(flat[o:o + n].view_as(p))
            o += n
