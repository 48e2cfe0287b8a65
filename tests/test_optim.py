"""Row a26: the optimiser set-up of d3h/optim.py against torch.optim.lr_scheduler.LambdaLR and the reference's grouping rules
(train.py:569-620 init, :862-912 split, :1273-1312 seq)."""
import os
import sys
import types

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'd3human-code_amd'))


class _Geo(torch.nn.Module):
    """parameter names of HmSDFTetsGeometry's state_dict (geometry/hmsdf.py:178-345)"""

    def __init__(self):
        super().__init__()
        P = lambda *s: torch.nn.Parameter(torch.randn(*s))
        self.sdf_net = torch.nn.Sequential(torch.nn.Linear(3, 4), torch.nn.Linear(4, 1))
        self.nonrigid = torch.nn.Linear(5, 3)
        self.sdf, self.msdf, self.deform = P(7), P(7), P(7, 3)
        self.cond, self.render_cond, self.fix_code = P(2, 64), P(2, 64), P(1, 1, 136)


def _flags():
    F = types.SimpleNamespace(learning_rate=[0.03, 0.005], use_msdf_mlp=False)
    F.trans_optim = torch.zeros(2, 3, requires_grad=True)
    for k in ('rhand_pose', 'jaw_pose', 'expr', 'body_pose', 'root_pose', 'lhand_pose', 'leye_pose', 'reye_pose'):
        setattr(F, k, torch.zeros(2, 3))
    return F


def test_pass_learning_rates_follow_train_py_indexing():
    from d3h import optim as O
    assert O.pass_learning_rates([0.03, 0.005], 0) == (0.03, 0.03, 0.03 * 6.0)          # the config's 0.005 is never used (train.py:569-572)
    assert O.pass_learning_rates([[0.03, 0.005, 0.1]], 0) == (0.03, 0.005, 0.1)
    assert O.pass_learning_rates(0.01) == (0.01, 0.01, 0.06)


def test_groups_match_the_reference_rules():
    from d3h import optim as O
    g, F = _Geo(), _flags()
    name = {id(p): n for n, p in g.named_parameters()}
    name[id(F.trans_optim)] = 'trans_optim'

    def table(stage):
        return [(round(grp['lr'] / 0.03, 6), sorted(name.get(id(p), 'pose') for p in grp['params'])) for grp in O.geometry_groups(stage, g, F, 0.03)]
    init = table('init')
    assert init[0] == (1e-3, ['trans_optim']) and all(lr == 1e-3 and ps == ['pose'] for lr, ps in init[1:9])
    assert init[9] == (1.0, ['deform'])
    assert init[10] == (1e-2, ['sdf', 'sdf_net.0.bias', 'sdf_net.0.weight', 'sdf_net.1.bias', 'sdf_net.1.weight'])       # no msdf in the init stage
    assert init[11] == (1e-3, ['cond', 'fix_code', 'render_cond'])
    split = table('split')
    assert split == [(1.0, ['deform']), (1.0, ['msdf']), (1e-3, ['nonrigid.bias', 'nonrigid.weight']), (1e-2, ['cond', 'fix_code', 'render_cond'])]
    seq = table('seq')
    assert seq == [(1e-2, ['nonrigid.bias', 'nonrigid.weight']), (1e-2, ['cond', 'render_cond'])]


def test_lambda_lr_equals_torch_lambda_lr_and_updates_are_identical():
    from d3h import optim as O
    torch.manual_seed(0)
    ga, gb = _Geo(), _Geo()
    gb.load_state_dict(ga.state_dict())
    Fa, Fb = _flags(), _flags()
    ma, mb = [torch.nn.Parameter(torch.ones(4))], [torch.nn.Parameter(torch.ones(4))]
    oa = O.make_optimizers('init', ga, ma, Fa, warmup_iter=300)
    ob = O.make_optimizers('init', gb, mb, Fb, warmup_iter=300,
                           scheduler_cls=lambda opt, fn: torch.optim.lr_scheduler.LambdaLR(opt, lr_lambda=lambda x: fn(x)))
    gen = torch.Generator().manual_seed(1)
    for it in range(420):
        for (geo, F, mat, (og, om, sch)) in ((ga, Fa, ma, oa), (gb, Fb, mb, ob)):
            og.zero_grad(); om.zero_grad()
        grads = [torch.randn(p.shape, generator=gen) for p in ga.parameters()]
        for geo, mat, (og, om, sch) in ((ga, ma, oa), (gb, mb, ob)):
            for p, gr in zip(geo.parameters(), grads):
                p.grad = gr.clone()
            mat[0].grad = torch.full((4,), 0.5)
            om.step(); sch[1].step()
            og.step(); sch[0].step()
        la = [grp['lr'] for grp in oa[0].param_groups] + [grp['lr'] for grp in oa[1].param_groups]
        lb = [grp['lr'] for grp in ob[0].param_groups] + [grp['lr'] for grp in ob[1].param_groups]
        assert all(abs(x - y) <= 1e-12 * max(1.0, abs(y)) for x, y in zip(la, lb)), (it, la, lb)
    for pa, pb in zip(ga.parameters(), gb.parameters()):
        assert torch.equal(pa, pb)
    assert torch.equal(ma[0], mb[0])
    assert abs(oa[0].param_groups[9]['lr'] - 0.03 * 10 ** (-(420 - 300) * 0.0002)) < 1e-12
