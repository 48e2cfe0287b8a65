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
