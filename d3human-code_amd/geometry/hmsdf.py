"""HmSDFTetsGeometry with the reference's interface for the init stage (geometry/hmsdf.py:178-273,310-345,382-413,416-523,706-737,
810-915): tet grid + SDF MLP + mSDF/deform parameters + SMPL-X deformer; getMesh_init / render_init / tick_init, on the MI355X
kernels.  Parameter names match the reference's state_dict (sdf_net.net.{i}.*, msdf, deform, sdf, cond, render_cond, fix_code).

Differences, all stated in DESIGN.md:
 * N-frame batches are real: every frame of target['idx'] is posed with its own SMPL-X parameters (the reference poses the whole
   batch with idx[0], hmsdf.py:471).  One SDF sweep / marching-tets / nearest-vertex search is shared by the batch.
 * The perceptual (MobileNetV2) normal loss needs torchvision weights that cannot be obtained offline; `normal_loss` is the
   reference's own MSE + 0.1 (1 - cos) formula (hmsdf.py:1067-1068) unless a `normal_loss_fn` is supplied.
 * FLAGS extensions (all optional): tet_grid=(verts, indices), smplx_model_dict, sdf_init_fn (analytic SDF for the pre-fit instead
   of pysdf), ssim_weight, render_buffers / render_buffers_split / render_buffers_seq ('all' = the reference's 12 buffers in tick_*;
   default: the buffers the tick reads, see _tick_buffers).
getMesh_split / render_split / tick_split (hmsdf.py:526-630,740-774,917-1096; hmSDF_Tets with type in {"cloth","body"}) follow the same
pattern.  The seq stage (getMesh_seq, tick_seq, render_mask) is the next row of SURVEY §8(f).
"""
import os

import numpy as np
import torch
from d3h._lib import cur_stream as _cur_stream
from d3h import _lib as _L
import torch.nn.functional as F

from render import mesh
from render import render
from render import regularizer
from render import renderutils as ru
import render.optixutils as ou
from d3h import imgops as _I
from d3h import mtets as _M
from d3h import gradarena as _GA
from deform.smplx_exavatar_deformer import SMPLX_Deformer
from .gshell_tets import GShell_Tets
from .hmsdf_tets_split import hmSDF_Tets
from .mlp import MLP, MLP_deform

AHEAD_EIK = os.environ.get('D3H_AHEAD_EIK', '1') != '0'      # '0': the ahead launches stop at the surface samples; the eikonal chain starts once the host is back
AHEAD = os.environ.get('D3H_LAUNCH_AHEAD', '1') != '0'      # '0': nearest vertex / LBS / sampler / first eikonal sweep only after the sizes are known (A/B)


def compute_sdf_reg_loss(sdf, all_edges, marks=None):
    """hmsdf.py:162-170 (all_edges: int64 [N_e,2] as the reference, or the int32 copy); marks: see d3h.imgops.sdf_reg_loss"""
    e32 = all_edges if all_edges.dtype == torch.int32 else all_edges.int()
    return _I.sdf_reg_loss(sdf, e32, marks)


def _flag(FLAGS, name, default=None):
    return getattr(FLAGS, name, default)




from .perceptual import MobileNetPerceptualLoss      # hmsdf.py:137-159 (torchvision-compatible trunk, see geometry/perceptual.py)


def crop_image(image1, image2, h, w, crop_size):
    """hmsdf.py:68-76: the same random crop_size^2 window of two [..., H, W] images.  Python's `random`, width offset drawn first, and
    (h, w) are what the caller passes -- FLAGS.texture_res in tick_split (hmsdf.py:1072), not the image size -- exactly as the reference,
    so a seeded run crops the same window; like the reference it raises (randint on an empty range) when crop_size > h or w."""
    import random
    start_w = random.randint(0, w - crop_size)
    start_h = random.randint(0, h - crop_size)
    return (image1[..., start_h:start_h + crop_size, start_w:start_w + crop_size],
            image2[..., start_h:start_h + crop_size, start_w:start_w + crop_size])


def collision_loss(cloth_pos, body_pos, body_faces, push_eps=0.005):
    """hmsdf.py:98-132: mean relu(push_eps - (p - c_f) . n_f)^2 over cloth vertices p, f = body face with the nearest centre c_f"""
    from d3h import meshops as _MO
    return _MO.collision_loss(cloth_pos, body_pos, body_faces, push_eps=push_eps)


class _PendingEikonal:
    """an eikonal term whose forward sweep is queued on the side stream and whose remaining launches are still to be issued"""
    __slots__ = ('pts', 'iteration', 'cus', 'begun')

    def __init__(self, pts, iteration, cus, begun):
        self.pts, self.iteration, self.cus, self.begun = pts, iteration, cus, begun


class _Ahead:
    """d3h/mtets.py: spec_hook of one extraction.  When the extraction runs speculatively, everything that needs its VERTICES only is queued by
    launch() at the capacity of the vertex buffer, with the row count read on the device, BEFORE the host knows the sizes: nearest SMPL-X vertex,
    LBS, the surface samples and the first sweep of the eikonal chain.  HmSDFTetsGeometry._extract: pose() then only narrows the results and builds
    the autograd nodes (tools/dbg/gpu_host_window.py: ~170 us of launches leave the stretch in which the GPU waits for the host)."""
    __slots__ = ('geo', 'target', 'posed', 'early', 'launched', 'ok', 'nn', 'flat', 'd', 'rnd')

    def __init__(self, geo, target, posed, early):
        self.geo, self.target, self.posed, self.early = geo, target, posed, early
        self.launched = self.ok = False
        self.nn = self.flat = self.rnd = None
        self.d = {}

    def launch(self, verts_cap, faces_cap, counts):
        geo, target = self.geo, self.target
        if target is None or verts_cap.shape[0] == 0 or not AHEAD:
            return
        d = geo.smplx_deform
        self.nn = d.nearest_counted(verts_cap, counts)
        self.flat = d.lbs_forward_counted(verts_cap, counts, self.nn, self.posed['transforms'])
        if self.early and faces_cap.shape[0] > 0:
            # frame 0 of the dense [frames, rows, 3] result = the leading rows of the flat buffer, whatever `rows` turns out to be.
            # The uniform numbers are drawn here and kept: an extraction that outgrows its capacity repeats the sampling on the SAME draws
            self.rnd = torch.rand(int(_flag(geo.FLAGS, 'eikonal_samples', 50000)), 3, device=verts_cap.device)
            geo._launch_eikonal(self.d, None, target, v0=self.flat[:verts_cap.shape[0] * 3].view(-1, 3), faces=faces_cap, begin_only=True,
                                rnd=self.rnd)
        self.launched = True


class HmSDFTetsGeometry(torch.nn.Module):
    # per-iteration bookkeeping attributes (plain Python values, never Parameters / Modules / buffers): written straight into __dict__ --
    # nn.Module.__setattr__ costs ~7 us a piece and a tick sets nine of them on the launch-bound part of the iteration
    _PLAIN = frozenset(('_sweep_cache', '_eik_it', '_eik_cus', '_eik_pending', '_side_stream', '_tick_skips_watertight', '_heads',
                        'last_mesh_dict', 'last_lpips_loss'))

    def __setattr__(self, name, value):
        if name in HmSDFTetsGeometry._PLAIN:
            self.__dict__[name] = value
        else:
            super().__setattr__(name, value)

    def __init__(self, grid_res, scale, FLAGS, offset=None):
        super().__init__()
        self.FLAGS, self.grid_res, self.scale = FLAGS, grid_res, scale
        self.gshell_tets, self.hmsdf_tets = GShell_Tets(), hmSDF_Tets()
        self.batch_point_num = 100000
        self.device = torch.device(_flag(FLAGS, 'device', 'cuda'))
        self.boxscale = torch.tensor(_flag(FLAGS, 'boxscale', [1, 1, 1]), dtype=torch.float32, device=self.device).view(1, 3)
        self.smplx_deform = SMPLX_Deformer(model_path='smplx', gender=_flag(FLAGS, 'gender', 'neutral'),
                                           model_dict=_flag(FLAGS, 'smplx_model_dict'), device=self.device,
                                           shape_param_dim=FLAGS.shape_param.shape[-1], expr_param_dim=FLAGS.expr_optim.shape[-1])
        self._init_tet()
        self._init_sdf()
        self._init_use_nonrigid_deform()
        self._init_msdf()
        self._init_deform()
        n_img = _flag(FLAGS, 'n_images', 1)
        self._init_cond(n_img)
        self._init_render_cond(n_img)
        self.fix_code = torch.nn.Parameter(0.1 * torch.randn((1, 1, 136), device=self.device), requires_grad=True)
        # hmsdf.py:190 builds MobileNetPerceptualLoss() unconditionally (downloads pretrained torchvision weights).  Here it is built
        # when a checkpoint is named (FLAGS.mobilenet_weights) or explicitly requested; tick_* then use it as the normal loss with the
        # reference's factors (50 / 5 on a 448^2 crop / 20), otherwise they use the MSE + cosine formula of hmsdf.py:1067-1068.
        if _flag(FLAGS, 'mobilenet_weights') is not None or _flag(FLAGS, 'use_perceptual_normal_loss', False):
            self.mobileNet_perceptual_loss = MobileNetPerceptualLoss(use_gpu=self.device.type == 'cuda', weights=_flag(FLAGS, 'mobilenet_weights'))
            if _flag(FLAGS, 'normal_loss_fn') is None:
                FLAGS.normal_loss_fn = self.mobileNet_perceptual_loss

    # ---- initialisation -------------------------------------------------------------------------------------------
    def _init_tet(self):
        with torch.no_grad():
            self.optix_ctx = ou.OptiXContext()
            grid = _flag(self.FLAGS, 'tet_grid')
            if grid is None:
                tets = np.load('data/tets/tet_grid.npz')                      # hmsdf.py:207
                v, idx = tets['vertices'], tets['indices']
                v = np.asarray(v, np.float32).copy()
                v[:, 1] -= 0.1919                                            # hmsdf.py:210-211
                v *= 1.2
            else:
                v, idx = grid                                                # synthetic grids already carry the offset/scale
            self.verts = torch.as_tensor(np.asarray(v), dtype=torch.float32, device=self.device).contiguous()
            self.indices = torch.as_tensor(np.asarray(idx), dtype=torch.long, device=self.device).contiguous()
            self.generate_edges()

    @torch.no_grad()
    def generate_edges(self):
        g = _M.TetGrid.get(self.indices)            # static per-grid data shared with the marching-tets kernels
        self.all_edges = g.all_edges                # == unique(sort(indices[:, edges])) (hmsdf.py:384-387)
        self.all_edges32 = g.edges32
        self.max_displacement = 1.0 / self.grid_res * self.scale / 2.1

    def _init_sdf(self):
        F_ = self.FLAGS
        self.sdf = torch.nn.Parameter(torch.zeros_like(self.verts[:, 0]), requires_grad=True)     # placeholder, as the reference
        self.sdf_net = MLP(skip_in=_flag(F_, 'skip_in', [3]), n_freq=_flag(F_, 'n_freq', 6), n_hidden=_flag(F_, 'n_hidden', 6),
                           d_hidden=_flag(F_, 'd_hidden', 256), use_float16=_flag(F_, 'use_float16', False)).to(self.device)
        self.smplx_deform.initialize(betas=F_.shape_param.to(self.device))
        steps = _flag(F_, 'sdf_mlp_pretrain_smpl_steps', 3000)
        ckp = None
        if _flag(F_, 'out_dir'):
            os.makedirs(os.path.join(F_.out_dir, 'ckp'), exist_ok=True)
            ckp = os.path.join(F_.out_dir, 'ckp', 'init_smpl_deform_convex_{}.pth'.format(self.grid_res))
        if ckp and os.path.exists(ckp):
            self.sdf_net.load_state_dict(torch.load(ckp, map_location=self.device))
            return
        fn = _flag(F_, 'sdf_init_fn')
        if fn is None:
            # hmsdf.py:236-237: sdf_gt = -pysdf.SDF(template verts, faces)(grid verts).  Here: the same quantity (exact distance,
            # containment sign, positive outside) from the brute-force GPU kernel d3h_mesh_sdf -- no CPU third party at start-up.
            from d3h import meshops as _MO
            if self.smplx_deform.layer.faces is None:
                raise RuntimeError('HmSDFTetsGeometry: the SDF pre-fit needs a body model with faces (the SMPL-X template mesh, '
                                   'hmsdf.py:232-237) or FLAGS.sdf_init_fn (an analytic target, as the synthetic scenes use)')
            faces = torch.as_tensor(np.asarray(self.smplx_deform.layer.faces).astype(np.int64), device=self.device)
            sdf_gt = _MO.mesh_sdf(self.verts, self.smplx_deform.vs_template[0].detach(), faces).reshape(-1, 1)
        else:
            sdf_gt = fn(self.verts).reshape(-1, 1).to(self.device)
        if steps > 0:
            opt = torch.optim.Adam(self.sdf_net.parameters(), lr=1e-3)           # hmsdf.py:254-271
            net = self.sdf_net.forward_reference if _flag(F_, 'prefit_with_library_path', False) else self.sdf_net
            for _ in range(steps):
                loss = (net(self.verts) - sdf_gt).pow(2).mean()
                opt.zero_grad()
                loss.backward()
                opt.step()
            self.sdf_prefit_loss = float(loss.detach())
        if ckp:
            torch.save(self.sdf_net.state_dict(), ckp)

    def _init_use_nonrigid_deform(self):
        if not _flag(self.FLAGS, 'use_nonrigid_deform', False):
            return
        self.nonrigid = MLP_deform(skip_in=_flag(self.FLAGS, 'skip_in', [3]), n_freq=8, n_hidden=_flag(self.FLAGS, 'n_hidden', 6),
                                   d_hidden=_flag(self.FLAGS, 'd_hidden', 256), d_out=3).to(self.device)
        self._load_or_pretrain_offset_net(self.nonrigid)

    def _load_or_pretrain_offset_net(self, net):
        """hmsdf.py:278-308: load checkpoints/init_deform_deform_cond_pe8.pth when present, otherwise fit the network to a zero
        offset on the grid vertices (Adam 1e-3, FLAGS.sdf_deform_pretrain_steps steps, zero pose code) and save it there"""
        path = _flag(self.FLAGS, 'deform_checkpoint', 'checkpoints/init_deform_deform_cond_pe8.pth')
        if path and os.path.exists(path):
            net.load_state_dict(torch.load(path, map_location=self.device))
            return
        steps = _flag(self.FLAGS, 'sdf_deform_pretrain_steps', 0)
        if steps <= 0:
            return
        code = torch.zeros(1, 1, 136, device=self.device)
        opt = torch.optim.Adam(net.parameters(), lr=1e-3)
        x = self.verts.reshape(1, -1, 3)
        fwd = net.forward_reference if _flag(self.FLAGS, 'prefit_with_library_path', False) else net
        for _ in range(steps):
            loss = fwd(x, code).pow(2).mean()
            opt.zero_grad()
            loss.backward()
            opt.step()
        self.offset_prefit_loss = float(loss.detach())
        if path and _flag(self.FLAGS, 'save_deform_checkpoint', False):
            os.makedirs(os.path.dirname(path) or '.', exist_ok=True)
            torch.save(net.state_dict(), path)

    def _init_use_body_nonrigid_deform(self):
        """hmsdf.py:343-380: a second MLP_deform (`body_nonrigid`), same construction / checkpoint as `nonrigid`; the seq stage's
        optimiser picks it up through the 'nonrigid' name filter (train.py:1296) although getMesh_seq only evaluates `nonrigid`"""
        if not _flag(self.FLAGS, 'use_nonrigid_deform', False):
            return
        self.body_nonrigid = MLP_deform(skip_in=_flag(self.FLAGS, 'skip_in', [3]), n_freq=8, n_hidden=_flag(self.FLAGS, 'n_hidden', 6),
                                        d_hidden=_flag(self.FLAGS, 'd_hidden', 256), d_out=3).to(self.device)
        self._load_or_pretrain_offset_net(self.body_nonrigid)

    def _init_msdf(self):
        msdf = (torch.rand_like(self.verts[:, 0]) - 0.01).clamp(-1, 1)          # hmsdf.py:311
        self.msdf = torch.nn.Parameter(msdf.clone().detach(), requires_grad=True)

    def _init_deform(self):
        self.deform = torch.nn.Parameter(torch.zeros_like(self.verts), requires_grad=True)
        self.clamp_deform()

    def _init_basedeform(self, v, f, body_v=None, cloth_v=None):
        self.base_v, self.base_f = v, f
        if body_v is not None:
            self.body_v = body_v
        if cloth_v is not None:
            self.cloth_v = cloth_v

    def _init_cond(self, img_num):
        self.cond = torch.nn.Parameter(torch.rand((img_num + 1, 64), device=self.device), requires_grad=True)

    def _init_render_cond(self, img_num):
        self.render_cond = torch.nn.Parameter(torch.rand((img_num + 1, 64), device=self.device), requires_grad=True)

    @torch.no_grad()
    def getAABB(self):
        return torch.min(self.verts, dim=0).values, torch.max(self.verts, dim=0).values

    @torch.no_grad()
    def clamp_deform(self):
        if not _flag(self.FLAGS, 'use_tanh_deform', False):
            self.deform.data.clamp_(-1.0, 1.0)            # in place: the same values as `data[:] = clamp(...)` (hmsdf.py:401-405), one launch
        self.msdf.data.clamp_(-2.0, 2.0)

    # ---- mesh extraction ---------------------------------------------------------------------------------------------
    def _smplx_param(self):
        F_ = self.FLAGS
        return {"shape": F_.shape_param, "face_offset": _flag(F_, 'face_offset'), "joint_offset": _flag(F_, 'joint_offset'),
                "locator_offset": _flag(F_, 'locator_offset'), "trans": F_.trans_optim, "rhand_pose": _flag(F_, 'rhand_pose_optim'),
                "jaw_pose": F_.jaw_pose_optim, "expr": F_.expr_optim, "body_pose": F_.body_pose_optim, "root_pose": F_.root_pose_optim,
                "lhand_pose": _flag(F_, 'lhand_pose_optim'), "leye_pose": _flag(F_, 'leye_pose_optim'), "reye_pose": _flag(F_, 'reye_pose_optim')}

    def _sdf_sweep(self):
        """hmsdf.py:433-444: v_deformed = verts + max_displacement * deform; sdf = sdf_net(v_deformed) -- one fused kernel sweep.
        Frame-parallel runs (FLAGS.sdf_shard = (rank, world)) evaluate 1/world of the grid per rank and all-gather (d3h.dist_ops)."""
        v_deformed = _GA.displace(self.verts, self.deform, self.max_displacement)      # verts + max_displacement * deform
        # FLAGS.share_sdf_sweep: the split stage extracts the garment and the body from the same network within one iteration
        # (tick_split x2 before one backward, train.py:1035-1100); the second call reuses the first sweep (and its graph) as long as no
        # parameter changed.  The caller drops `_sweep_cache` at the start of every iteration (Scene.step_split).
        share = _flag(self.FLAGS, 'share_sdf_sweep', False) and _flag(self.FLAGS, 'use_sdf_mlp', True)
        if share:
            key = (tuple(p._version for p in self.sdf_net.parameters()), self.deform._version, torch.is_grad_enabled())
            hit = getattr(self, '_sweep_cache', None)
            if hit is not None and hit[0] == key:
                return v_deformed, hit[1]
        if _flag(self.FLAGS, 'use_sdf_mlp', True):
            sh = _flag(self.FLAGS, 'sdf_shard')
            pk = self._tick_pack = self.sdf_net.pack() if self.sdf_net.fused else None      # reused by the eikonal term of this tick
            if sh is not None and sh[1] >= 1:          # (world 1 is the whole grid through the same row-range / gather path: the single-rank RCCL test)
                from d3h import dist_ops as _D
                n = self.verts.shape[0]
                lo, hi, shard = _D.shard_range(n, sh[0], sh[1])
                if self.sdf_net.fused:
                    loc = self.sdf_net(self.verts, deform=self.deform, disp=self.max_displacement, pack=pk, rows=(lo, hi))
                else:
                    loc = self.sdf_net(v_deformed[lo:hi])
                sdf = _D.gather_shards(loc, n, shard, sh[0], sh[1])
            elif self.sdf_net.fused:
                sdf = self.sdf_net(self.verts, deform=self.deform, disp=self.max_displacement, pack=pk)
            else:
                sdf = self.sdf_net(v_deformed)
        else:
            sdf = self.sdf
        if share:
            self._sweep_cache = (key, sdf)
        return v_deformed, sdf

    def _extract(self, material, target, tets_fn):
        """shared body of getMesh_init / getMesh_split (hmsdf.py:416-523 / 526-630)"""
        v_deformed, sdf = self._sdf_sweep()
        if torch.is_grad_enabled() and sdf.requires_grad:
            # The first half of the sweep's compact backward -- the list of grid vertices on sign-changing edges (the only ones d(loss)/d(sdf)
            # can be non-zero at: marching tets interpolates between them, the regulariser penalises those edges), their gather and the
            # recompute of their activations: three launches of the step's serial tail -- is queued NOW on a stream of its own: the stretch
            # that follows (extraction, LBS, sampler: ~20 small dependent launches) leaves the chip idle (d3h.sdf_mlp.prepare_backward)
            from d3h import sdf_mlp as _SM
            _SM.prepare_backward(sdf, self.all_edges32)
        msdf = self.msdf
        want_wt = self._want_watertight()
        posed = {}
        early = os.environ.get('D3H_EARLY_EIKONAL', '1') != '0'

        ahead = _Ahead(self, target, posed, early)
        if target is not None:
            # (the frame transforms depend on the pose parameters only: computed before the extraction -- the host is ahead of the GPU here --
            # with autograd on, for the speculative launches and for pose() alike)
            posed['frames'] = list(target['idx']) if isinstance(target['idx'], (list, tuple)) else [int(target['idx'])]
            posed['param'] = self._smplx_param()
            posed['transforms'] = self.smplx_deform.frame_transforms(posed['param'], posed['frames'])

        def pose(verts, verts_wt, faces_padded=None):
            # Everything that needs the extracted VERTICES only -- nearest SMPL-X vertex + LBS of the mesh and of its watertight twin --
            # is queued here, before marching tets reads the cut-face count back: that host sync then waits behind these kernels instead
            # of leaving the GPU idle (d3h/mtets.py).
            if target is None:
                return
            frames, param = posed['frames'], posed['param']
            if ahead.ok and verts.shape[0] > 0:
                from d3h import lbs as _HL
                p_ = verts.shape[0]
                posed['verts'] = self.smplx_deform.lbs_forward_batch(verts, param, frames, nn_idx=ahead.nn[:p_], transforms=posed['transforms'],
                                                                     pre=_HL.counted_result(ahead.flat, len(frames), p_))
                if 'sampled_pts' in ahead.d:
                    posed['sampled_pts'], posed['_eik'] = ahead.d['sampled_pts'], ahead.d.get('_eik')
                    if '_eik_deferred' in ahead.d and posed['sampled_pts'] is not None:
                        posed['_eik'] = self._eikonal_async(posed['sampled_pts'], *ahead.d['_eik_deferred'])
                    faces_padded = None                  # (the samples are drawn, the eikonal chain is running)
            else:
                nn_idx = self.smplx_deform.nearest(verts) if verts.shape[0] > 0 else None
                posed['verts'] = self.smplx_deform.lbs_forward_batch(verts, param, frames, nn_idx=nn_idx, transforms=posed['transforms']) \
                    if verts.shape[0] > 0 else verts.new_zeros(len(frames), 0, 3)
            # The eikonal chain is the longest dependency chain of the step and it needs only surface SAMPLES: drawn here from the face list at
            # its allocation bound (zero-area padding rows have probability exactly 0: same samples from the same random stream), its first sweep is
            # queued BEFORE the host blocks in the cut-face read-back and builds the mesh objects -- ~0.15 ms earlier on a GPU that is
            # otherwise idle in that stretch (profiles/r4_bench_config3_timeline.csv: 390 us of idle gaps per iteration, all of them here).
            if faces_padded is not None and faces_padded.shape[0] > 0 and verts.shape[0] > 0 and early:
                self._launch_eikonal(posed, None, target, v0=posed['verts'][0], faces=faces_padded, rnd=getattr(ahead, 'rnd', None))
            if want_wt:
                # watertight vertices are the first n_wt rows of verts_aug wherever those are referenced; unreferenced rows of
                # verts_aug are zeroed (gshell_tets.py:423-427), so the search is repeated on the un-zeroed watertight set
                posed['wt'] = self.smplx_deform.lbs_forward_batch(verts_wt, param, frames, nn_idx=self.smplx_deform.nearest(verts_wt),
                                                                  transforms=posed['transforms']) \
                    if verts_wt.shape[0] > 0 else verts_wt.new_zeros(len(frames), 0, 3)

        verts, faces, uvs, uv_idx, v_tng, extra = tets_fn(v_deformed, sdf, msdf, self.indices, pose, ahead)
        f32, fwt32 = extra['faces32'], extra['faces_watertight32']
        ret = {}
        template_imesh = mesh.Mesh(verts, faces, material=material, t_pos_idx32=f32)
        imesh = mesh.auto_normals(template_imesh, lazy=True)       # canonical-space normals: computed if somebody reads them
        ret['tmp_nodeform_mesh'] = imesh            # identical content (the reference builds it twice, hmsdf.py:459-467,484-491)
        deform_imesh = None
        if target is not None:
            deform_imesh = mesh.auto_normals(mesh.Mesh(posed['verts'], faces, material=material, t_pos_idx32=f32), lazy=True)
            if 'sampled_pts' in posed and faces.shape[0] == 0:
                # the early launch sampled the zero-padded face list of a pass that extracted vertices but NO face (an empty garment / body in
                # a fresh split stage): every row was degenerate, the samples are meaningless -- as the reference, whose sampler is ill-defined
                # there, the term is skipped (the chain already queued on the side stream is dropped; nothing reads it)
                posed.pop('_eik', None)
                ret['sampled_pts'] = None
            elif 'sampled_pts' in posed:               # sampled and launched before the face read-back (pose() above)
                ret['sampled_pts'], ret['_eik'] = posed['sampled_pts'], posed.get('_eik')
            else:
                self._launch_eikonal(ret, deform_imesh, target)
        ret.update({'imesh': imesh, 'deform_imesh': deform_imesh, 'template_imesh': template_imesh, 'sdf': sdf, 'msdf': extra['msdf'],
                    'msdf_watertight': extra['msdf_watertight'], 'msdf_boundary': extra['msdf_boundary'],
                    'n_verts_watertight': extra['n_verts_watertight']})
        if want_wt:
            wt = mesh.Mesh(extra['vertices_watertight'], extra['faces_watertight'], material=material, t_pos_idx32=fwt32)
            imesh_wt = mesh.auto_normals(wt, lazy=True)
            if target is not None:
                ret['tmp_nodeform_wt_mesh'] = imesh_wt
                ret['deform_imesh_wt'] = mesh.auto_normals(mesh.Mesh(posed['wt'], extra['faces_watertight'], material=material,
                                                                     t_pos_idx32=fwt32), lazy=True)
            ret['imesh_watertight'] = imesh_wt
        return ret

    def getMesh_init(self, material, target=None, it=None):
        return self._extract(material, target, lambda p, s, m, t, early, ahead: self.gshell_tets(p, s, m, t, _before_face_sync=early, _spec_hook=ahead))

    def getMesh_split(self, material, type, target=None, it=None):
        return self._extract(material, target, lambda p, s, m, t, early, ahead: self.hmsdf_tets(p, s, m, t, type, _before_face_sync=early, _spec_hook=ahead))

    def _launch_eikonal(self, d, opt_mesh, target=None, v0=None, faces=None, begin_only=False, rnd=None):
        """Surface samples for the eikonal term (hmsdf.py:714,750) and the term itself, launched on the side stream as soon as the posed
        mesh exists: its chain of sweeps (forward, gradient, tangent, reverse, weight-gradient GEMMs: ~4 ms at 50 000 points) is the
        longest dependency chain of the forward phase, so it starts first; the watertight-mesh posing, both renders and the loss
        kernels overlap it."""
        import kaolin
        if v0 is None and opt_mesh is not None and opt_mesh.v_pos.shape[-2] != 0 and opt_mesh.t_pos_idx.shape[0] != 0:      # (no faces: sampler ill-defined)
            v0 = opt_mesh.v_pos[0] if opt_mesh.v_pos.dim() == 3 else opt_mesh.v_pos
            faces = opt_mesh.t_pos_idx
        if v0 is not None:
            with torch.no_grad():        # the only consumer (the eikonal term) detaches them (hmsdf.py:858)
                d['sampled_pts'] = kaolin.ops.mesh.sample_points(v0[None, ...], faces, _flag(self.FLAGS, 'eikonal_samples', 50000), _rnd=rnd)[0][0]      # 50000: hmsdf.py:714,750
        else:
            d['sampled_pts'] = None
        it = getattr(self, '_eik_it', None)
        if it is not None and d['sampled_pts'] is not None and _flag(self.FLAGS, 'use_sdf_mlp', True) and _flag(self.FLAGS, 'use_eikonal', True):
            pixels = 0
            if isinstance(target, dict) and 'resolution' in target and torch.is_tensor(target.get('mvp')):
                pixels = int(target['mvp'].shape[0]) * int(target['resolution'][0]) * int(target['resolution'][1]) * int(target.get('spp', 1)) ** 2
            if begin_only and not (AHEAD_EIK and d['sampled_pts'].is_cuda and self.sdf_net.fused and os.environ.get('D3H_EIK_SPLIT_ISSUE', '1') != '0'
                                   and os.environ.get('D3H_NO_SIDE_STREAM') != '1'):
                # called from inside an autograd Function (grad mode off): only the graph-free first sweep of the split chain may be queued
                # there; any other form of the term is evaluated by the caller once it is back in grad mode (_extract: pose)
                d['_eik_deferred'] = (it, pixels)
            else:
                d['_eik'] = self._eikonal_async(d['sampled_pts'], it, pixels)

    def _render(self, d, glctx, target, lgt, bsdf, denoiser, shadow_scale, use_uv, buffers, grad_buffers=None):
        opt_mesh, original_mesh = d['deform_imesh'], d['tmp_nodeform_mesh']
        if 'sampled_pts' not in d:
            self._launch_eikonal(d, opt_mesh, target)
        idx0 = target['idx'][0] if isinstance(target['idx'], (list, tuple)) else target['idx']
        d['buffers'] = render.render_mesh(self.FLAGS, idx0, glctx, opt_mesh, original_mesh, target['mvp'], target['campos'], lgt,
                                          target['resolution'], spp=target['spp'], msaa=True, background=target['background'], bsdf=bsdf,
                                          use_uv=use_uv, optix_ctx=self.optix_ctx, denoiser=denoiser, shadow_scale=shadow_scale,
                                          extra_dict={'msdf': d['msdf']}, buffers=buffers, _grad_buffers=grad_buffers)
        self._eikonal_finish(d)            # the rest of the eikonal chain, now that the render's forward launches are queued
        if self._want_watertight():
            with torch.no_grad():          # feeds no loss (hmsdf.py:729-735, train.py:1627): rendered for the validation images only
                d['buffers_watertight'] = render.render_mesh(self.FLAGS, idx0, glctx, d['deform_imesh_wt'], d['tmp_nodeform_wt_mesh'],
                                                             target['mvp'], target['campos'], lgt, target['resolution'], spp=target['spp'],
                                                             msaa=True, background=target['background'], bsdf=bsdf, use_uv=use_uv,
                                                             optix_ctx=self.optix_ctx, extra_dict=None, buffers=buffers)
        return d

    def render_init(self, glctx, target, lgt, opt_material, bsdf=None, denoiser=None, shadow_scale=1.0, use_uv=False, iteration=None,
                    buffers=None, grad_buffers=None):
        self._eik_it = iteration if _flag(self.FLAGS, '_want_eikonal', False) else None
        try:
            d = self.getMesh_init(opt_material, target=target, it=iteration)
        finally:
            self._eik_it = None
        return self._render(d, glctx, target, lgt, bsdf, denoiser, shadow_scale, use_uv, buffers, grad_buffers)

    def render_split(self, glctx, target, lgt, opt_material, type, bsdf=None, denoiser=None, shadow_scale=1.0, use_uv=False, iteration=None,
                     buffers=None, grad_buffers=None):
        self._eik_it = iteration if _flag(self.FLAGS, '_want_eikonal', False) else None
        try:
            d = self.getMesh_split(opt_material, type, target=target, it=iteration)
        finally:
            self._eik_it = None
        return self._render(d, glctx, target, lgt, bsdf, denoiser, shadow_scale, use_uv, buffers, grad_buffers)

    def _want_watertight(self):
        """FLAGS.visualize_watertight (hard-coded True, train.py:1627) makes getMesh_* / render_* pose and render the watertight twin of the
        mesh -- for the validation images: no loss reads `buffers_watertight` and tick_* return loss values only (hmsdf.py:729-735,810-915),
        so inside a tick the render is unobservable.  tick_* therefore skip it unless they are asked for the reference's full output
        (FLAGS.render_buffers* = 'all'); render_init / render_split called directly (validate_itr*, train.py:419-537) always honour the flag."""
        return _flag(self.FLAGS, 'visualize_watertight', False) and not getattr(self, '_tick_skips_watertight', False)

    # ---- losses ----------------------------------------------------------------------------------------------------------------
    def _tick_buffers(self, flag, reads):
        """Which buffers a tick_* asks render_mesh for.  The reference renders all 12 every time (render.py:430-449) although a tick
        reads three to seven of them and returns only loss values: nothing a caller of tick_* can observe depends on the others, so the
        default is `reads` (dead-output elimination).  FLAGS.<flag> = 'all' renders the full set as the reference does; a tuple names
        the set explicitly.  render_init / render_split / render_seq called directly (validate_itr*, train.py:419-537) keep the full
        set as their default."""
        want = _flag(self.FLAGS, flag)
        if want is None:
            return tuple(reads)
        if isinstance(want, str):
            if want != 'all':
                raise ValueError(f"FLAGS.{flag} = {want!r}: expected 'all', None or a tuple of buffer names")
            return None
        return tuple(want)

    def _eikonal(self, pts, iteration, begun=None, cus=0):
        """hmsdf.py:856-876; the gradient graph is the fused second-order op of d3h.sdf_mlp (MLP.input_gradient)"""
        es = _flag(self.FLAGS, 'eikonal_scale')
        if es is None:
            eik_coeff = 3e-1 if iteration < 500 else (1e-1 if iteration < 2000 else 1e-2)
        else:
            eik_coeff = es
        return self.sdf_net.eikonal_loss(pts, eik_coeff, pack=getattr(self, '_tick_pack', None), begun=begun, max_cus=cus) if self.sdf_net.fused else \
            self.sdf_net.eikonal_loss(pts, eik_coeff)

    def _eikonal_async(self, pts, iteration, pixels=0):
        """The eikonal branch depends only on the sampled surface points and the SDF weights, so on the GPU it is issued on a second
        HIP stream: its kernels (forward, gradient, tangent, reverse and weight-gradient sweeps over 50 000 points) overlap the render /
        loss kernels of the main stream, in the forward and -- because autograd replays every node on the stream it was recorded
        on -- in the backward as well."""
        if pts.shape[0] == 0:                          # FLAGS.eikonal_samples = 0: the mean over no samples -- the term is skipped
            return None
        if not pts.is_cuda or os.environ.get('D3H_NO_SIDE_STREAM') == '1':       # (profiling: serialised, every kernel timed alone)
            return self._eikonal(pts, iteration)
        main = _cur_stream()
        if getattr(self, '_side_stream', None) is None:
            self._side_stream = torch.cuda.Stream()
            try:        # the SDF weights are read on both streams on purpose; autograd syncs their gradient accumulation
                torch.autograd.graph.set_warn_on_accumulate_grad_stream_mismatch(False)
            except AttributeError:
                pass
        side = self._side_stream
        side.wait_stream(main)
        # The chain kernels fill a CU's register file, so render / loss kernels of the main stream only run on CUs the chain leaves
        # out.  Its 128-point tiles are spread evenly over the rounds the whole chip needs (50 000 samples: 391 tiles, 2 rounds, 196
        # CUs do what 256 would) and, when the main stream is heavy (>= 2 Mpixel per step) and the chain is longer than one round, over
        # one round more (131 CUs, 3 rounds; config 3: 7.48 against 7.62 ms per step, config 2 with its 1 Mpixel: 5.2 against 4.2 ms,
        # hence the condition).  The cap is an argument of every launch of the chain (C ABI: max_cus), not process state.
        # D3H_EIK_CUS=<n> overrides (clamped to 1..256), 0 = whole chip.
        ntiles = (int(pts.shape[0]) + 127) // 128
        rounds = -(-ntiles // 256)
        if pixels >= (2 << 20) and ntiles > 256:
            rounds += 1
        cus = -(-ntiles // rounds)
        env = os.environ.get('D3H_EIK_CUS')
        if env is not None:
            cus = min(256, max(0, int(env)))
        cus = cus if 0 < cus < 256 else 0
        if ntiles <= 128:
            cus = 0          # a launch below two tiles per CU is spread at wave granularity by the launch itself (sdf_mlp_layout.h)
        self._eik_cus = cus if cus else 256        # (bench.py reports the chain kernels' fraction of these CUs as well)
        split = self.sdf_net.fused and os.environ.get('D3H_EIK_SPLIT_ISSUE', '1') != '0'
        with _L.use_stream(side):
            if split:
                # Only the chain's first kernel (the forward sweep with the activation save, the longest launch of the chain) is
                # queued now; the caller issues the render's launches while it runs and _eikonal_finish queues the rest -- the host
                # needs ~0.15 ms for the chain's remaining launches, which otherwise delays the first render kernel by as much.
                begun = self.sdf_net.eikonal_begin(pts, pack=getattr(self, '_tick_pack', None), max_cus=cus)
                return _PendingEikonal(pts, iteration, cus, begun)
            e = self._eikonal(pts, iteration, cus=cus)
        self._eik_pending = side
        return e

    def _eikonal_finish(self, d):
        """second half of _eikonal_async (everything after the forward sweep), on the side stream; no-op when nothing is pending"""
        p = d.get('_eik')
        if not isinstance(p, _PendingEikonal):
            return
        side = self._side_stream
        cus = p.cus
        env = os.environ.get('D3H_EIK_CUS_FINISH')            # (experiment: another CU budget for the chain's kernels after its first sweep)
        if env is not None and cus:
            cus = min(256, max(1, int(env)))
        with _L.use_stream(side):
            d['_eik'] = self._eikonal(p.pts, p.iteration, begun=p.begun, cus=cus)
        self._eik_pending = side

    def _eikonal_join(self, e):
        side = getattr(self, '_eik_pending', None)
        if side is not None:
            ev = getattr(e, 'd3h_ready', None) if (self.sdf_net.fused and os.environ.get('D3H_EIK_FULL_JOIN') != '1') else None
            if ev is not None:
                # only the loss value is needed here; the eager second-order sweeps of the term keep running on the side stream under
                # the loss / render backward kernels of this stream (its backward node is replayed on the side stream, after them)
                _cur_stream().wait_event(ev)
            else:
                _cur_stream().wait_stream(side)
            e.record_stream(_cur_stream())
            self._eik_pending = None
        return e

    # ---- seq stage (hmsdf.py:632-704, 773-808, 1099-1182) -----------------------------------------------------------------------
    def _seq_index(self):
        """static per-run data of the seq stage: where cloth / body vertices sit in base_v (FLAGS.v_labels), int32 face lists"""
        c = getattr(self, '_seq_cache', None)
        vl = self.FLAGS.v_labels
        if c is None or c['src'] is not vl:
            is_cloth = (vl == 1)
            order = torch.cat([torch.nonzero(is_cloth).reshape(-1), torch.nonzero(vl == 0).reshape(-1)])
            inv = torch.empty_like(order)
            inv[order] = torch.arange(order.shape[0], device=order.device)
            c = self._seq_cache = {'src': vl, 'idx_cloth': torch.nonzero(is_cloth).reshape(-1), 'idx_body': torch.nonzero(vl == 0).reshape(-1),
                                   'gather': inv, 'n_cloth': int(is_cloth.sum()), 'covered': bool(((vl == 0) | (vl == 1)).all())}
        return c

    def getMesh_seq(self, material, target=None, it=None, save_tmp=False, t="all"):
        F_ = self.FLAGS
        f = self.base_f
        f32 = getattr(self, '_base_f32', None)
        if f32 is None or self._base_f32_src is not f:
            self._base_f32_src, self._base_f32 = f, f.int().contiguous()
            f32 = self._base_f32
        ret = {}
        if target is not None:
            sq = self._seq_index()
            # hmsdf.py:658-665: one pass of the non-rigid network over [cloth_v; body_v] (it is point-wise), scattered to base_v order
            both = self.nonrigid(torch.cat([self.cloth_v, self.body_v]).reshape(1, -1, 3), self.fix_code).reshape(-1, 3)
            if sq['covered']:
                delta = both.index_select(0, sq['gather'])          # (index_select: its backward is one index_add; `both[idx]` sorts the indices)
            else:
                delta = torch.zeros_like(self.base_v).index_put((torch.cat([sq['idx_cloth'], sq['idx_body']]),), both)
            delta_v = self.base_v + delta
            ret['delta'] = delta
            ret['tmp_nodeform_mesh'] = mesh.auto_normals(mesh.Mesh(self.base_v, f, material=material, v_labels=F_.v_labels,
                                                                  face_labels=F_.face_labels, t_pos_idx32=f32))
            if save_tmp:
                ret['tmp_all_mesh'] = mesh.auto_normals(mesh.Mesh(delta_v, f, material=material, t_pos_idx32=f32))
            frames = list(target['idx']) if isinstance(target['idx'], (list, tuple)) else [int(target['idx'])]
            v1 = self.smplx_deform.lbs_forward_batch(delta_v, self._smplx_param(), frames[:1], nn_idx=self.smplx_deform.nearest(delta_v))[0]
        else:
            v1 = self.base_v
        all_mesh = mesh.Mesh(v1, f, material=material, v_labels=F_.v_labels, face_labels=F_.face_labels,
                             connected_faces=_flag(F_, 'connected_faces'), edges=_flag(F_, 'edges'), t_pos_idx32=f32)
        ret['all_mesh'] = mesh.auto_normals(all_mesh)
        return ret

    def render_seq(self, glctx, target, lgt, opt_material, bsdf=None, denoiser=None, shadow_scale=1.0, use_uv=False, iteration=None, t="all",
                   buffers=None, grad_buffers=None):
        from render import render_mask
        d = self.getMesh_seq(opt_material, target=target, it=iteration, save_tmp=True, t=t)
        all_mesh = d['all_mesh']
        idx0 = target['idx'][0] if isinstance(target['idx'], (list, tuple)) else target['idx']
        d['all_mesh_buffers'] = render_mask.render_mesh(self.FLAGS, idx0, glctx, all_mesh, d['tmp_nodeform_mesh'], target['mvp'], target['campos'],
                                                        lgt, target['resolution'], spp=target['spp'], msaa=True, background=target['background'],
                                                        bsdf=bsdf, use_uv=use_uv, optix_ctx=self.optix_ctx, denoiser=denoiser,
                                                        shadow_scale=shadow_scale, buffers=buffers, _grad_buffers=grad_buffers)
        b = d['all_mesh_buffers']
        v_label_render = b['mesh_id'][..., 0]
        d['_label'] = v_label_render                                                 # (the fused loss pass of tick_seq forms the masks itself)
        alpha = b['geometric_normal'][..., -1]
        d['cloth_mask'] = v_label_render * alpha                                     # hmsdf.py:790-797
        d['body_mask'] = (1 - v_label_render) * alpha
        d['all_mask'] = alpha
        sq = self._seq_index()
        v = all_mesh.v_pos
        d['colli_loss'] = collision_loss(v.index_select(0, sq['idx_cloth']), v.index_select(0, sq['idx_body']), self.FLAGS.body_f)      # hmsdf.py:799-806
        return d

    def tick_seq(self, glctx, target, lgt, opt_material, loss_fn, iteration, denoiser=None, t="all"):
        """hmsdf.py:1099-1182; the caller weights the terms (train.py:1412-1421)"""
        from lap_loss import body_laplacian_loss, body_normal_loss
        F_ = self.FLAGS
        # ('_seen_faces': the per-triangle visibility bitmap.  `visible_triangles` of the result is its deferred compaction -- the loop reads it
        # once, after the last iteration (train.py:1515), and compacting it every iteration is a host synchronisation in mid-step)
        reads = ('shaded', 'geometric_normal', 'kd_grad', 'ks_grad', 'normal_grad', '_seen_faces') + \
            (('kd',) if _flag(F_, 'lambda_chroma', 0.0) != 0 else ())
        want = self._tick_buffers('render_buffers_seq', reads)
        d = self.render_seq(glctx, target, lgt, opt_material, use_uv=False, denoiser=denoiser, t=t, buffers=want, grad_buffers=reads)
        b = d['all_mesh_buffers']
        all_mesh = d['all_mesh']
        with torch.no_grad():
            gt_cloth, gt_body, gt_all, gt_all_normal = target['cloth_img'], target['body_img'], target['all_img'], target['all_normal']
        m_all, m_cloth, m_body = d['all_mask'][..., None], d['cloth_mask'][..., None], d['body_mask'][..., None]
        vis = b.get('visible_triangles')                  # a plain tensor when the render was asked for it (FLAGS.render_buffers_seq = 'all' / a tuple naming it)
        if vis is None and b.get('_seen_faces') is not None:
            vis = render.LazyVisibleTriangles(b['_seen_faces'])
        out = {'visible_triangles': vis, 'delta': d['delta']}
        rgb = b['shaded'][..., 0:3]
        st_, lay_ = b.get('_stacked'), b.get('_layout') or {}
        spec = ru.loss_spec(loss_fn, rgb.device) if st_ is not None else None
        if st_ is not None and spec is not None and 'shaded' in lay_ and 'geometric_normal' in lay_ and '_label' in d \
                and os.environ.get('D3H_SEQ_FUSED_TERMS', '1') != '0':
            # the three mask MSEs and the three image losses in one pass over the stacked render (d3h.imgops.seq_losses): masks alpha,
            # label * alpha, (1 - label) * alpha exactly as render_seq forms them
            from d3h import imgops as _Im
            sl = _Im.seq_losses(st_, lay_, d['_label'], gt_all, gt_cloth, gt_body, spec) * self._const((200.0, 200.0, 200.0, 1.0, 1.0, 1.0), rgb.device)
            out['all_msk_loss'], out['cloth_msk_loss'], out['body_msk_loss'] = sl[0], sl[1], sl[2]
            out['all_img_loss'], out['cloth_img_loss'], out['body_img_loss'] = sl[3], sl[4], sl[5]
        else:
            out['all_msk_loss'] = 200 * F.mse_loss(m_all, gt_all[..., 3:])
            out['cloth_msk_loss'] = 200 * F.mse_loss(m_cloth, gt_cloth[..., 3:])
            out['body_msk_loss'] = 200 * F.mse_loss(m_body, gt_body[..., 3:])
            out['all_img_loss'] = loss_fn(rgb * m_all, gt_all[..., 0:3])
            out['cloth_img_loss'] = loss_fn(rgb * m_cloth, gt_cloth[..., 0:3])
            out['body_img_loss'] = loss_fn(rgb * m_body, gt_body[..., 0:3])
        # material smoothness and the normal term in the fused per-pixel pass of tick_init / tick_split (the same formulas: regularizer.py:47-52,
        # hmsdf.py:1067-1068) when the buffers are this build's stacked render: ~25 elementwise / reduction launches over the image less
        nfn = _flag(F_, 'normal_loss_fn')
        st, layout = b.get('_stacked'), b.get('_layout') or {}
        fused_terms = None
        if st is not None and nfn is None and all(k in layout for k in ('geometric_normal', 'kd_grad', 'ks_grad', 'normal_grad')) \
                and os.environ.get('D3H_SEQ_FUSED_TERMS', '1') != '0':
            from d3h import imgops as _Im
            pl = _Im.pixel_losses(st, {k: v for k, v in layout.items() if k in ('geometric_normal', 'kd_grad', 'ks_grad', 'normal_grad')}, gt_all,
                                  gt_all_normal[..., 0:3], None, False)
            fused_terms = pl['vec'] * self._const((1.0, 1.0, 0.5, 0.5, 1.0, 1.0, float(_flag(F_, 'lambda_kd', 0.1)), float(_flag(F_, 'lambda_ks', 0.05)),
                                                   float(_flag(F_, 'lambda_nrm', 0.025)), 1.0), rgb.device)
        if fused_terms is not None:
            mtl = fused_terms[6:9].sum()
        else:
            mtl = regularizer.material_smoothness_grad(b['kd_grad'], b['ks_grad'], b['normal_grad'], lambda_kd=_flag(F_, 'lambda_kd', 0.1),
                                                       lambda_ks=_flag(F_, 'lambda_ks', 0.05), lambda_nrm=_flag(F_, 'lambda_nrm', 0.025))
        lam_c = _flag(F_, 'lambda_chroma', 0.0)          # 0 in the reference's configuration (train.py:1598): the term is mean(...) * 0
        chroma = regularizer.chroma_loss(b['kd'], gt_all, lam_c) if lam_c != 0 else torch.zeros((), device=rgb.device)
        out['mtl_smooth_loss'], out['chroma_loss'] = mtl, chroma
        out['shading_reg_loss'] = out['reg_loss'] = mtl + chroma
        out['delta_loss'] = torch.sum(torch.norm(d['delta'], dim=1) ** 2)
        if fused_terms is not None:
            out['normal_loss'] = fused_terms[4] + 0.1 * (1 - fused_terms[5])
        else:
            out_n = F.normalize(b['geometric_normal'][..., 0:3], p=2, dim=-1) * self._const((1.0, -1.0, -1.0), rgb.device)
            gt_n = F.normalize(gt_all_normal[..., 0:3], p=2, dim=-1)
            if nfn is not None:           # reference: 20 x MobileNetV2 feature L1 on the [0,1]-mapped normal images (hmsdf.py:1150-1154)
                out['normal_loss'] = 20 * nfn(((out_n + 1) / 2).permute(0, 3, 1, 2), ((gt_n + 1) / 2).permute(0, 3, 1, 2))
            else:                         # no pretrained trunk offline: the MSE + cosine formulation of hmsdf.py:1067-1068
                out['normal_loss'] = F.mse_loss(out_n, gt_n) + 0.1 * (1 - F.cosine_similarity(out_n.reshape(-1, 3), gt_n.reshape(-1, 3), dim=1).mean())
        out['laplacian_loss'] = body_laplacian_loss(all_mesh)
        out['nds_normal_loss'] = body_normal_loss(all_mesh)
        out['colli_loss'] = d['colli_loss']
        self.last_mesh_dict = d
        return out

    def _const(self, vals, dev):
        c = getattr(self, '_const_cache', None)
        if c is None:
            c = self._const_cache = {}
        k = (vals, str(dev))
        if k not in c:
            c[k] = torch.tensor(vals, dtype=torch.float32, device=dev)
        return c[k]

    def _fused_pixel_vec(self, buffers, color_ref, normal_ref, loss_fn, want_ssim):
        """raw output vector of the fused per-pixel loss pass (d3h.imgops.PIXEL_LOSS_KEYS order), or None when that pass does not
        apply (foreign buffers, a loss_fn without `d3h_spec`, the perceptual normal loss)"""
        st, layout = buffers.get('_stacked'), buffers.get('_layout')
        spec = ru.loss_spec(loss_fn, st.device) if st is not None else None
        if st is None or 'shaded' not in layout or spec is None or _flag(self.FLAGS, 'normal_loss_fn') is not None:
            return None
        from d3h import imgops as _I
        has_n = 'geometric_normal' in buffers and normal_ref is not None
        pl = _I.pixel_losses(st, layout, color_ref, normal_ref[..., 0:3] if has_n else None, spec, want_ssim)
        return pl['vec'], layout, has_n

    def _pixel_terms(self, buffers, color_ref, normal_ref, loss_fn, want_ssim, masked_prep=None):
        """The per-pixel loss terms shared by tick_init and tick_split (hmsdf.py:835-839,895-898 / 969-975,1064-1068): mask MSE, image
        loss + the two msdf_image L1 terms, normal MSE / cosine, optional SSIM.  One fused pass over render_mesh's stacked output
        (d3h.imgops.pixel_losses) when the buffers come from this build's render_mesh and `loss_fn` declares its (loss, tonemapper)
        through a `d3h_spec` attribute; the same formulas as separate torch ops otherwise (foreign buffers, perceptual normal loss)."""
        gt_mask = color_ref[..., 3:]
        dev = color_ref.device
        has_n = 'geometric_normal' in buffers and normal_ref is not None
        perceptual = _flag(self.FLAGS, 'normal_loss_fn') is not None
        st, layout = buffers.get('_stacked'), buffers.get('_layout')
        spec = ru.loss_spec(loss_fn, st.device) if st is not None else None
        out = {'normal_mse': None, 'normal_cos': None, 'ssim': None, 'out_n': None, 'gt_n': None, 'mtl_smooth': None, 'masked': None}
        if st is not None and 'shaded' in layout:
            from d3h import imgops as _I
            pl = _I.pixel_losses(st, layout, color_ref, normal_ref[..., 0:3] if (has_n and not perceptual) else None, spec, want_ssim,
                                 masked_prep=masked_prep)
            out['masked'] = pl['masked']       # shaded.rgb * ref.a as the LPIPS trunk input, out of the same pass (None unless asked for)
            # one fused multiply over the 7 means instead of a dozen scalar kernels (and as many autograd nodes): the iteration is
            # host-bound in this stretch.  t = [mask, img, .5 msdf+, .5 msdf-, normal mse, normal cos, kd_grad, ks_grad, normal_grad, ssim]
            F_ = self.FLAGS
            t = pl['vec'] * self._const((1.0, 1.0, 0.5, 0.5, 1.0, 1.0, float(_flag(F_, 'lambda_kd', 0.1)), float(_flag(F_, 'lambda_ks', 0.05)),
                                         float(_flag(F_, 'lambda_nrm', 0.025)), 1.0), dev)
            if all(k in layout for k in ('kd_grad', 'ks_grad', 'normal_grad')):
                out['mtl_smooth'] = t[6:9].sum()          # regularizer.material_smoothness_grad (render/regularizer.py:47-52)
            out['mask_mse'] = t[0]
            if spec is not None:
                out['img'] = t[1:4].sum() if 'msdf_image' in layout else t[1]
            else:
                img = loss_fn(buffers['shaded'][..., 0:3] * gt_mask, color_ref[..., 0:3] * gt_mask)
                out['img'] = img + t[2:4].sum() if 'msdf_image' in layout else img
            if has_n and not perceptual:
                out['normal_mse'], out['normal_cos'] = t[4], t[5]
            if want_ssim:
                out['ssim'] = t[9]
        else:
            out['mask_mse'] = F.mse_loss(buffers['shaded'][..., 3:], color_ref[..., 3:])
            img = loss_fn(buffers['shaded'][..., 0:3] * gt_mask, color_ref[..., 0:3] * gt_mask)
            if 'msdf_image' in buffers:
                mi = buffers['msdf_image']
                img = img + 5e-1 * F.l1_loss(mi.clamp(min=0) * (gt_mask == 0).float(), torch.zeros_like(gt_mask))
                img = img + 5e-1 * F.l1_loss(mi.clamp(max=0) * (gt_mask == 1).float(), torch.ones_like(gt_mask))
            out['img'] = img
            if want_ssim:
                import ssim_loss
                a = (buffers['shaded'][..., 0:3] * gt_mask).permute(0, 3, 1, 2)
                b = (color_ref[..., 0:3] * gt_mask).permute(0, 3, 1, 2)
                out['ssim'] = ssim_loss.ssim(a.contiguous(), b.contiguous())
        if has_n and out['normal_mse'] is None:
            out_n = F.normalize(buffers['geometric_normal'][..., 0:3], p=2, dim=-1) * self._const((1.0, -1.0, -1.0), dev)
            gt_n = F.normalize(normal_ref[..., 0:3], p=2, dim=-1)
            out['out_n'], out['gt_n'] = out_n, gt_n
            out['normal_mse'] = F.mse_loss(out_n, gt_n)
            out['normal_cos'] = F.cosine_similarity(out_n.reshape(-1, 3), gt_n.reshape(-1, 3), dim=1).mean()
        return out

    def tick_init(self, glctx, target, lgt, opt_material, loss_fn, iteration, denoiser=None):
        F_ = self.FLAGS
        t_iter = iteration / F_.iter
        shadow_ramp = min(iteration / 1000, 1.0)
        reads = ('shaded', 'geometric_normal', 'msdf_image')                                                # hmsdf.py:835-839,895
        want = self._tick_buffers('render_buffers', reads)
        F_._want_eikonal = True
        self._tick_skips_watertight = want is not None                # (the reference-equivalent 'all' mode renders the watertight twin too)
        try:
            d = self.render_init(glctx, target, lgt, opt_material, denoiser=denoiser, shadow_scale=shadow_ramp, iteration=iteration, buffers=want,
                                 grad_buffers=reads)
        finally:
            F_._want_eikonal = False
            self._tick_skips_watertight = False
        buffers = d['buffers']
        color_ref = target['all_img']
        gt_mask = color_ref[..., 3:]
        zero = torch.zeros((), device=color_ref.device)

        sw = _flag(F_, 'ssim_weight', 0.0)
        fused = self._fused_pixel_vec(buffers, color_ref, target.get('all_normal'), loss_fn, want_ssim=bool(sw))
        self._eikonal_finish(d)
        eik_loss = self._eikonal_join(d['_eik']) if d.get('_eik') is not None else zero      # after the pixel pass is enqueued: overlap
        sdf_weight = F_.sdf_regularizer - (F_.sdf_regularizer - 0.01) * min(1.0, 4.0 * t_iter)             # hmsdf.py:881
        sdf_reg = compute_sdf_reg_loss(d['sdf'], self.all_edges32)
        sdf_reg_loss = (sdf_reg if sdf_reg.dim() == 0 else sdf_reg.mean()) * sdf_weight
        if fused is not None:
            # every remaining combination is affine in the raw terms: one matrix-vector product (d3h/losshead.py) instead of ~25
            # scalar kernels and as many autograd nodes
            vec, layout, has_n = fused
            key = ('init', 'msdf_image' in layout, has_n, float(sw), str(vec.device))
            head = self._heads.get(key) if hasattr(self, '_heads') else None
            if head is None:
                if not hasattr(self, '_heads'):
                    self._heads = {}
                img = {1: 1.0, 2: 0.5, 3: 0.5} if 'msdf_image' in layout else {1: 1.0}                    # hmsdf.py:836-839
                msk = {0: 100.0}                                                                          # hmsdf.py:835
                nrm = ({4: 1.0, 5: -0.1}, 0.1) if has_n else ({}, 0.0)          # MSE + 0.1 (1 - cos) (hmsdf.py:1067-1068 on :895-898)
                ssm = ({9: -float(sw)}, float(sw)) if sw else ({}, 0.0)          # sw (1 - SSIM) (ssim_loss.py:33)
                tot = {10: 1.0, 11: 1.0}
                for coef in (msk, nrm[0], ssm[0]):
                    for j, v in coef.items():
                        tot[j] = tot.get(j, 0.0) + v
                rows = {'img_loss': (img, 0.0), 'msk_loss': (msk, 0.0), 'normal_loss': nrm, 'ssim_loss': ssm, 'sdf_reg_loss': ({10: 1.0}, 0.0),
                        'eik_loss': ({11: 1.0}, 0.0), 'reg_loss': ({10: 1.0, 11: 1.0}, 0.0),
                        'd3h_total': (tot, nrm[1] + ssm[1])}                                              # train.py:718 (+ the SSIM term)
                from d3h.losshead import AffineHead
                head = self._heads[key] = AffineHead(rows, 12, vec.device)
            h = head(torch.cat([vec, sdf_reg_loss.reshape(1), eik_loss.reshape(1)]))
            out = {"img_loss": h['img_loss'], "depth_loss": zero, "sdf_reg_loss": h['sdf_reg_loss'], "eik_loss": h['eik_loss'],
                   "msk_loss": h['msk_loss'], "delta_loss": zero, "reg_loss": h['reg_loss'], "geo_reg_loss": h['reg_loss'],
                   "normal_loss": h['normal_loss'], "d3h_total": h['d3h_total']}
            if sw:
                out['ssim_loss'] = h['ssim_loss']
            self.last_mesh_dict = d
            return out

        px = self._pixel_terms(buffers, color_ref, target.get('all_normal'), loss_fn, want_ssim=bool(sw))
        msk_loss = 100 * px['mask_mse']                                                                    # hmsdf.py:835
        img_loss = px['img']                                                                              # hmsdf.py:836-839
        geo_reg_loss = sdf_reg_loss + eik_loss
        reg_loss = geo_reg_loss

        # normal term: reference formula hmsdf.py:895-898 for the unit normals, then MSE + 0.1 (1 - cos) (hmsdf.py:1067-1068)
        if px['normal_mse'] is not None:
            nfn = _flag(F_, 'normal_loss_fn')
            if nfn is not None:
                normal_loss = 50 * nfn(((px['out_n'] + 1) / 2).permute(0, 3, 1, 2), ((px['gt_n'] + 1) / 2).permute(0, 3, 1, 2))
            else:
                normal_loss = px['normal_mse'] + 0.1 * (1 - px['normal_cos'])
        else:
            normal_loss = zero

        out = {"img_loss": img_loss, "depth_loss": zero, "sdf_reg_loss": sdf_reg_loss, "eik_loss": eik_loss, "msk_loss": msk_loss,
               "delta_loss": zero, "reg_loss": reg_loss, "geo_reg_loss": geo_reg_loss, "normal_loss": normal_loss}
        if sw:                                              # BASELINE config 3: (1 - SSIM(shaded, all_img)) -- ssim_loss.py:33
            out['ssim_loss'] = sw * (1.0 - px['ssim'])
        self.last_mesh_dict = d
        return out

    def tick_split(self, glctx, target, lgt, opt_material, loss_fn, iteration, denoiser=None, type='cloth'):
        """hmsdf.py:917-1096.  Totals are assembled by the caller (train.py:1050-1087: img + normal + reg + 10 * msk per garment/body)."""
        F_ = self.FLAGS
        t_iter = iteration / F_.iter
        shadow_ramp = min(iteration / 1000, 1.0)
        reads = ('shaded', 'geometric_normal', 'msdf_image', 'kd_grad', 'ks_grad', 'normal_grad') + \
            (('kd',) if _flag(F_, 'lambda_chroma', 0.0) != 0 else ())                                       # hmsdf.py:947-951,1036-1043,1062
        want = self._tick_buffers('render_buffers_split', reads)
        if want is not None and 'visible_triangles' not in want and '_seen_faces' not in want:
            want = tuple(want) + ('_seen_faces',)                     # read by the mesh-mSDF regulariser below (hmsdf.py:1010-1017)
        F_._want_eikonal = True
        self._tick_skips_watertight = _flag(F_, 'render_buffers_split') is None or not isinstance(_flag(F_, 'render_buffers_split'), str)
        try:
            d = self.render_split(glctx, target, lgt, opt_material, type, denoiser=denoiser, shadow_scale=shadow_ramp, iteration=iteration,
                                  buffers=want, grad_buffers=reads)
        finally:
            F_._want_eikonal = False
            self._tick_skips_watertight = False
        buffers = d['buffers']
        key = {'cloth': 'cloth', 'body': 'body', 'all': 'all'}[type]
        color_ref, normal_ref = target[key + '_img'], target[key + '_normal']
        gt_mask = color_ref[..., 3:]
        dev = color_ref.device
        zero = torch.zeros((), device=dev)

        lp = _flag(F_, 'lpips_fn')
        # the LPIPS input ((2 (shaded.rgb * mask) - 1) - shift) / scale comes out of the fused per-pixel pass, and its gradient goes back in
        # through that pass: no masking / normalising / scaling kernels over the image, no slice of the stacked image with its zero-filled
        # gradient (D3H_LPIPS_FUSED_INPUT=0: the separate torch ops)
        prep = lp.prepare_constants() if (lp is not None and hasattr(lp, 'prepare_constants')
                                          and os.environ.get('D3H_LPIPS_FUSED_INPUT', '1') != '0') else None
        px = self._pixel_terms(buffers, color_ref, normal_ref, loss_fn, want_ssim=False, masked_prep=prep)
        msk_loss = px['mask_mse']
        img_loss = px['img']

        # LPIPS on the masked colour images (extension: BASELINE config 5 names LPIPS in the full loss stack; the reference vendors the
        # package, third_parties/lpips, but no training path calls it).  FLAGS.lpips_fn: an lpips.LPIPS module; added to img_loss.
        lpips_loss = None
        if lp is not None:
            pre = px['masked'] is not None
            a = px['masked'].permute(0, 3, 1, 2) if pre else (buffers['shaded'][..., 0:3] * gt_mask).permute(0, 3, 1, 2)
            kw = {'in0_prepared': True} if pre else {}
            # the target side of the distance is a constant of the frame: its trunk features are computed once per target TENSOR OBJECT
            # (a third of the LPIPS convolutions).  An entry holds the tensor itself, so neither its id nor its storage can be re-used by
            # another tensor while it is cached, and an in-place change shows in `_version`; a loader that builds new tensors every
            # iteration simply misses.
            if hasattr(lp, 'reference_features') and not color_ref.requires_grad:
                cache = self.__dict__.setdefault('_lpips_ref_cache', {})
                ck = (type, id(color_ref))
                hit = cache.get(ck)
                if hit is None or hit[0] is not color_ref or hit[1] != color_ref._version:
                    if len(cache) >= 8:
                        cache.clear()
                    hit = cache[ck] = (color_ref, color_ref._version,
                                       lp.reference_features((color_ref[..., 0:3] * gt_mask).permute(0, 3, 1, 2).float()))
                lpips_loss = lp(a, None, ref_features=hit[2], **kw).mean() * _flag(F_, 'lpips_weight', 1.0)
            else:
                b = (color_ref[..., 0:3] * gt_mask).permute(0, 3, 1, 2).float()
                lpips_loss = lp(a, b, **kw).mean() * _flag(F_, 'lpips_weight', 1.0)
            img_loss = img_loss + lpips_loss

        self._eikonal_finish(d)
        eik_loss = self._eikonal_join(d['_eik']) if d.get('_eik') is not None else zero

        if _flag(F_, 'use_mesh_msdf_reg', True):                                           # hmsdf.py:996-1028
            regscale = (64 / self.grid_res) ** 3
            eps = self._const((1e-3,), dev)
            open_scale, close_scale = _flag(F_, 'msdf_reg_open_scale', 1e-6), _flag(F_, 'msdf_reg_close_scale', 3e-6)
            mesh_msdf_reg_loss = zero
            if open_scale > 0:
                m = d['msdf']
                mesh_msdf_reg_loss = open_scale * regscale * F.huber_loss(m.clamp(min=-eps).reshape(-1), -eps.expand(m.shape[0]), reduction='sum')
            if close_scale != 0:
                # boundary vertices that belong to a visible triangle (hmsdf.py:1010-1021).  The reference compacts them with a boolean
                # index (a host synchronisation); the masked sum below is the same value without leaving the stream
                with torch.no_grad():
                    n_wt = d['n_verts_watertight']
                    nb = d['msdf_boundary'].shape[0]
                    tp = d['imesh'].t_pos_idx
                    seen = buffers.get('_seen_faces')
                    if seen is None:
                        seen = torch.zeros(tp.shape[0], dtype=torch.bool, device=dev)
                        seen.index_fill_(0, buffers['visible_triangles'], True)
                    cnt = torch.zeros(n_wt + nb, dtype=torch.float32, device=dev)
                    cnt.index_add_(0, tp.reshape(-1), seen.float()[:, None].expand(-1, 3).reshape(-1))
                    visible_boundary = (cnt[n_wt:] > 0).float()
                bm = d['msdf_boundary'].clamp(max=eps).reshape(-1)
                mesh_msdf_reg_loss = mesh_msdf_reg_loss + close_scale * regscale * (
                    F.huber_loss(bm, eps.expand(bm.shape[0]), reduction='none') * visible_boundary).sum()
        else:
            mesh_msdf_reg_loss = zero

        sdf_weight = F_.sdf_regularizer - (F_.sdf_regularizer - 0.01) * min(1.0, 4.0 * t_iter)
        sdf_reg_loss = compute_sdf_reg_loss(d['sdf'], self.all_edges32).mean() * sdf_weight
        monochrome_loss = torch.zeros_like(img_loss)                                       # no 'diffuse_light' under bsdf = 'kd'
        if px['mtl_smooth'] is not None:              # evaluated inside the fused per-pixel pass
            mtl_smooth_loss = px['mtl_smooth']
        else:
            mtl_smooth_loss = regularizer.material_smoothness_grad(buffers['kd_grad'], buffers['ks_grad'], buffers['normal_grad'],
                                                                   lambda_kd=_flag(F_, 'lambda_kd', 0.1), lambda_ks=_flag(F_, 'lambda_ks', 0.05),
                                                                   lambda_nrm=_flag(F_, 'lambda_nrm', 0.025))
        lam_c = _flag(F_, 'lambda_chroma', 0.0)
        # lambda_chroma = 0 in the reference's configuration (train.py:1598): the term is mean(...) * 0; skip the passes over the image
        chroma_loss = regularizer.chroma_loss(buffers['kd'], color_ref, lam_c) if lam_c != 0 else zero
        geo_reg_loss = sdf_reg_loss + eik_loss
        shading_reg_loss = monochrome_loss + mtl_smooth_loss + chroma_loss
        reg_loss = geo_reg_loss + shading_reg_loss

        normal_loss_mse = px['normal_mse']                                                 # hmsdf.py:1067-1068
        normal_loss_cos = 0.1 * (1 - px['normal_cos'])
        nfn = _flag(F_, 'normal_loss_fn')
        if nfn is not None:         # reference: 5 x MobileNetV2 feature loss on a random 448^2 crop (hmsdf.py:1069-1074)
            a, b = ((px['out_n'] + 1) / 2).permute(0, 3, 1, 2), ((px['gt_n'] + 1) / 2).permute(0, 3, 1, 2)
            tr = _flag(F_, 'texture_res')           # hmsdf.py:1072 passes FLAGS.texture_res as (h, w); without the flag: the image size,
            th, tw = (int(tr[0]), int(tr[1])) if tr is not None else (a.shape[-2], a.shape[-1])      # crop clamped to it (no reference case)
            a, b = crop_image(a, b, th, tw, crop_size=448 if tr is not None else min(448, th, tw))
            normal_loss = 5 * nfn(a, b)
        else:
            normal_loss = normal_loss_mse + normal_loss_cos
        self.last_mesh_dict = d
        if lpips_loss is not None:
            self.last_lpips_loss = lpips_loss.detach()
        return {"img_loss": img_loss, "msk_loss": msk_loss, "depth_loss": zero, "sdf_reg_loss": sdf_reg_loss, "eik_loss": eik_loss,
                "mesh_msdf_reg_loss": mesh_msdf_reg_loss, "monochrome_loss": monochrome_loss, "mtl_smooth_loss": mtl_smooth_loss,
                "chroma_loss": chroma_loss, "delta_loss": zero, "reg_loss": reg_loss, "geo_reg_loss": geo_reg_loss,
                "shading_reg_loss": shading_reg_loss, "normal_loss_mse": normal_loss_mse, "normal_loss_cos": normal_loss_cos,
                "normal_loss": normal_loss}
