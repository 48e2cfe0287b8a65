"""The optimiser set-up of the reference's three training stages (train.py:569-620 init, :862-912 split, :1273-1312 seq), as data:
which parameters go in which Adam group at which learning rate, and the LambdaLR schedule -- so that Scene.step*() (bench.py, the
tests) updates exactly what train.py would.

Reference quirks kept on purpose (none of them is "fixed" here):
 * `FLAGS.learning_rate = [0.03, 0.005]` (configs/*.json) is indexed by pass_idx FIRST (train.py:569): with pass_idx = 0 every stage sees
   the scalar 0.03 for positions AND material, and 0.18 for the light; the 0.005 is never used.
 * The init stage optimises deform / the SDF network / the pose translation; `msdf` is in no group there (train.py:601-614), and the
   non-`_optim` pose tensors it lists have no gradient.
 * The split stage optimises deform, msdf, non-rigid and "other" parameters -- NOT the SDF network (train.py:896-902), although
   its gradient is still computed by backward().
 * Adam eps = 1e-8 is spelled out for the geometry optimisers and for the split / seq material optimisers; the init-stage material
   optimiser uses the default (the same value).
"""
import torch


def pass_learning_rates(learning_rate, pass_idx=0):
    """train.py:569-572 -> (lr_pos, lr_mat, lr_lgt)"""
    lr = learning_rate[pass_idx] if isinstance(learning_rate, (list, tuple)) else learning_rate
    is_seq = isinstance(lr, (list, tuple))
    return (lr[0] if is_seq else lr), (lr[1] if is_seq else lr), (lr[2] if is_seq else lr * 6.0)


def lr_schedule(warmup_iter):
    """train.py:573-576: linear warm-up, then 10^(-0.0002 (it - warmup))"""
    def fn(it):
        if it < warmup_iter:
            return it / warmup_iter
        return max(0.0, 10 ** (-(it - warmup_iter) * 0.0002))
    return fn


def _named(geometry, pred):
    return [p for n, p in geometry.named_parameters() if pred(n)]


def geometry_groups(stage, geometry, FLAGS, lr_pos):
    """param groups of optimizer_mesh in the reference's order"""
    if stage == 'init':                                                                    # train.py:591-614
        pose = [getattr(FLAGS, k, None) for k in ('trans_optim', 'rhand_pose', 'jaw_pose', 'expr', 'body_pose', 'root_pose', 'lhand_pose',
                                                  'leye_pose', 'reye_pose')]
        groups = [{'params': [t], 'lr': lr_pos * 1e-3} for t in pose if torch.is_tensor(t)]
        groups += [{'params': _named(geometry, lambda n: 'deform' in n), 'lr': lr_pos},
                   {'params': _named(geometry, lambda n: 'sdf' in n and 'msdf' not in n and 'smpl_msdf' not in n), 'lr': lr_pos * 1e-2},
                   {'params': _named(geometry, lambda n: 'sdf' not in n and 'msdf' not in n and 'deform' not in n and 'nonrigid' not in n
                                     and 'smpl_msdf' not in n), 'lr': lr_pos * 1e-3}]
    elif stage == 'split':                                                                 # train.py:887-902
        lr_msdf = lr_pos * 1e-2 if getattr(FLAGS, 'use_msdf_mlp', False) else lr_pos
        groups = [{'params': _named(geometry, lambda n: 'deform' in n), 'lr': lr_pos},
                  {'params': _named(geometry, lambda n: 'msdf' in n), 'lr': lr_msdf},
                  {'params': _named(geometry, lambda n: 'nonrigid' in n), 'lr': lr_pos * 1e-3},
                  {'params': _named(geometry, lambda n: 'sdf' not in n and 'msdf' not in n and 'deform' not in n and 'nonrigid' not in n
                                    and 'smpl_msdf' not in n), 'lr': lr_pos * 1e-2}]
    elif stage == 'seq':                                                                   # train.py:1295-1303
        groups = [{'params': _named(geometry, lambda n: 'nonrigid' in n), 'lr': lr_pos * 1e-2},
                  {'params': _named(geometry, lambda n: 'cond' in n), 'lr': lr_pos * 1e-2}]
    else:
        raise ValueError(stage)
    return [g for g in groups if len(g['params']) > 0]


class LambdaLR:
    """lr_k = base_lr * lr_lambda(k) per param group -- the closed form torch.optim.lr_scheduler.LambdaLR evaluates (pinned against it in
    tests/test_optim.py), without its per-step Python bookkeeping (0.35 ms per iteration for two schedulers on the launch-bound tail
    of the step)"""

    def __init__(self, optimizer, lr_lambda):
        self.opt, self.fn, self.k = optimizer, lr_lambda, 0
        self.base = [g['lr'] for g in optimizer.param_groups]
        self._apply()

    def _apply(self):
        f = self.fn(self.k)
        for g, b in zip(self.opt.param_groups, self.base):
            g['lr'] = b * f

    def step(self):
        self.k += 1
        self._apply()

    def get_last_lr(self):
        return [g['lr'] for g in self.opt.param_groups]


def make_optimizers(stage, geometry, material_params, FLAGS, warmup_iter=300, pass_idx=0, fused=False, scheduler_cls=LambdaLR):
    """-> (optimizer_mesh, optimizer_material, [scheduler_mesh, scheduler_material]) as train.py builds them for `stage`"""
    lr_pos, lr_mat, _ = pass_learning_rates(FLAGS.learning_rate, pass_idx)
    sched = lr_schedule(warmup_iter)
    opt_geo = torch.optim.Adam(geometry_groups(stage, geometry, FLAGS, lr_pos), eps=1e-8, fused=fused)
    opt_mat = torch.optim.Adam([{'params': list(material_params), 'lr': lr_mat}], eps=1e-8, fused=fused)
    return opt_geo, opt_mat, [scheduler_cls(opt_geo, sched), scheduler_cls(opt_mat, sched)]


class FusedAdam:
    """torch.optim.Adam (eps 1e-8, betas (0.9, 0.999), no weight decay / amsgrad) over every parameter of BOTH optimisers of a training
    stage in one kernel launch (csrc/optim.hip: d3h_adam_multi), with the per-tensor extras the loop applies around the steps folded
    in: a gradient scale (train.py:747-748: encoder gradient / 8) and a clamp of the updated values (geometry.clamp_deform,
    hmsdf.py:398-405; torch.clamp semantics, NaN propagates).  Parameters whose .grad is None are skipped, as torch does.  `param_groups` has torch's layout ('params', 'lr'),
    so the LambdaLR above drives it unchanged.  Pinned against torch.optim.Adam in tests/test_optim.py / the GPU parity tests."""

    def __init__(self, param_groups, betas=(0.9, 0.999), eps=1e-8):
        self.param_groups = [dict(g, params=list(g['params'])) for g in param_groups]
        self.betas, self.eps = betas, eps
        self.state = {}                       # id(param) -> [m, v, step]
        self._gscale, self._clamp = {}, {}

    def set_grad_scale(self, param, scale):
        self._gscale[id(param)] = float(scale)

    def set_clamp(self, param, lo, hi):
        self._clamp[id(param)] = (float(lo), float(hi))

    def zero_grad(self, set_to_none=True):
        for g in self.param_groups:
            for p in g['params']:
                if p.grad is not None:
                    if set_to_none:
                        p.grad = None
                    else:
                        p.grad.zero_()

    @torch.no_grad()
    def step(self):
        import ctypes
        from . import _lib as L
        # The argument arrays are built once per set of parameters-with-gradients and then only refreshed where a step changes them
        # (gradient pointers, learning rates, step counts): this call sits on the launch-bound tail of the iteration, where every
        # 10 us of host time is GPU idle time (profiles/r4_bench_config3_timeline.csv).
        live = []
        for g in self.param_groups:
            lr = g['lr']
            for p in g['params']:
                if p.grad is not None:
                    live.append((p, lr))
        nt = len(live)
        if nt == 0:
            return
        key = tuple(id(p) for p, _ in live)
        c = self.__dict__.get('_args')
        if c is None or c['key'] != key or c['gs_clamp'] != (self._gscale, self._clamp) or any(p.data_ptr() != a for (p, _), a in zip(live, c['P'])) \
                or any(self.state.get(id(p)) is not st_ for (p, _), st_ in zip(live, c['state'])):
            for p, _ in live:
                if id(p) not in self.state:
                    self.state[id(p)] = [torch.zeros_like(p, memory_format=torch.contiguous_format),
                                         torch.zeros_like(p, memory_format=torch.contiguous_format), 0]
                if not p.is_contiguous():
                    raise RuntimeError('FusedAdam: parameters must be contiguous')
                if not L.emulated() and not p.is_cuda:
                    raise RuntimeError('d3h: tensors must live on the GPU (the product has no CPU path)')
            st = [self.state[id(p)] for p, _ in live]
            inf = float('inf')
            c = self._args = {
                'key': key, 'plist': [p for p, _ in live], 'state': st,
                'P': (ctypes.c_void_p * nt)(*[p.data_ptr() for p, _ in live]), 'G': (ctypes.c_void_p * nt)(),
                'M': (ctypes.c_void_p * nt)(*[s_[0].data_ptr() for s_ in st]), 'V': (ctypes.c_void_p * nt)(*[s_[1].data_ptr() for s_ in st]),
                'N': (ctypes.c_int64 * nt)(*[p.numel() for p, _ in live]), 'LR': (ctypes.c_float * nt)(), 'ST': (ctypes.c_int64 * nt)(),
                'GS': (ctypes.c_float * nt)(*[self._gscale.get(id(p), 1.0) for p, _ in live]),
                'LO': (ctypes.c_float * nt)(*[self._clamp.get(id(p), (-inf, inf))[0] for p, _ in live]),
                'HI': (ctypes.c_float * nt)(*[self._clamp.get(id(p), (-inf, inf))[1] for p, _ in live]),
                'gs_clamp': (dict(self._gscale), dict(self._clamp))}
        G, LR, ST, keep = c['G'], c['LR'], c['ST'], []
        M, V = c['M'], c['V']
        # the kernel takes raw pointers: a gradient of another dtype / device / size (a foreign producer or hook) would be reinterpreted as
        # float* without a word -- three attribute compares per tensor (~0.3 us), BEFORE any step counter moves
        for i, (p, _) in enumerate(live):
            gr = p.grad
            if gr.dtype is not torch.float32 or gr.device != p.device or gr.numel() != p.numel():
                raise RuntimeError(f'd3h FusedAdam: gradient #{i} is {gr.dtype} on {gr.device} with {gr.numel()} elements; the parameter is '
                                   f'float32 on {p.device} with {p.numel()}')
        for i, ((p, lr), st) in enumerate(zip(live, c['state'])):
            M[i], V[i] = st[0].data_ptr(), st[1].data_ptr()      # (moment tensors may be swapped in place by a checkpoint restore)
            gr = p.grad
            if not gr.is_contiguous():
                gr = gr.contiguous()
                keep.append(gr)
            st[2] += 1
            G[i], LR[i], ST[i] = gr.data_ptr(), lr, st[2]
        L.check(L.lib().d3h_adam_multi(c['P'], G, c['M'], c['V'], c['N'], LR, ST, c['GS'], c['LO'], c['HI'], L.i32(nt), L.f32(self.betas[0]),
                                       L.f32(self.betas[1]), L.f32(self.eps), L.stream()), 'adam_multi')
        del keep
        # the kernel wrote through raw pointers: tell autograd (saved-tensor checks) and every cache keyed on `_version` (weight packs,
        # the shared SDF sweep, the deformer's transforms) that the parameters changed, as an in-place torch op would have
        torch.autograd.graph.increment_version(c['plist'])


def make_fused_optimizer(stage, geometry, material, FLAGS, warmup_iter=300, pass_idx=0):
    """ONE FusedAdam holding the groups of optimizer_mesh and of the material optimiser of `stage` (same groups, learning rates and
    schedule as make_optimizers), the encoder-gradient scale and the deform / msdf clamps -> (optimizer, scheduler)"""
    lr_pos, lr_mat, _ = pass_learning_rates(FLAGS.learning_rate, pass_idx)
    groups = geometry_groups(stage, geometry, FLAGS, lr_pos) + [{'params': list(material.parameters()), 'lr': lr_mat}]
    opt = FusedAdam(groups)
    enc = getattr(getattr(material, 'encoder', None), 'params', None)
    if enc is not None:
        opt.set_grad_scale(enc, 1.0 / 8.0)                                   # train.py:747-748
    if hasattr(geometry, 'deform') and not getattr(FLAGS, 'use_tanh_deform', False):
        opt.set_clamp(geometry.deform, -1.0, 1.0)                            # hmsdf.py:401-403
    if hasattr(geometry, 'msdf'):
        opt.set_clamp(geometry.msdf, -2.0, 2.0)                              # hmsdf.py:405
    return opt, LambdaLR(opt, lr_schedule(warmup_iter))
