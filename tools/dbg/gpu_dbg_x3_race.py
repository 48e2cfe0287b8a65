"""Two ticks of one scene state (plain path, then the frame-parallel arena path with the whole grid as rank 0's shard, as in
tests/test_gpu_parity.py::test_gpu_rccl_collectives_single_rank) compared tensor by tensor, repeated: hunting an intermittent difference.
    python tools/dbg/gpu_dbg_x3_race.py [repeats]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'd3human-code_amd')]
import torch
from d3h.scene import Scene
from d3h import gradarena, dist_ops
import torch.distributed as dist
os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29533')
if os.environ.get('NO_PG') != '1':
    dist.init_process_group('nccl', rank=0, world_size=1)
torch.manual_seed(0)
ell = lambda x: (((x - torch.tensor([0.0, -0.4, 0.0], device=x.device)) / torch.tensor([0.55, 0.8, 0.45], device=x.device)).norm(dim=-1) - 1.0) * 0.4
RES, GRID, FR = int(os.environ.get('RES', 128)), int(os.environ.get('GRID', 12)), int(os.environ.get('FRAMES', 2))
sc = Scene(res=RES, grid_n=GRID, n_frames=FR, device='cuda', prefit_steps=150, loss_set='full', body_verts=2048, sdf_fn=ell) if GRID <= 16 else \
    Scene(res=RES, grid_n=GRID, n_frames=FR, device='cuda', prefit_steps=300, loss_set='full', visualize_watertight=True)
bg = torch.rand(FR, RES, RES, 3, device='cuda')
names = ['deform', 'table'] + [n for n, _ in sc.geometry.sdf_net.named_parameters()] + ['h:msdf', 'h:posed', 'h:sdf', 'h:verts']

def tick(parallel):
    sc.world = 2 if parallel else 1
    sc.FLAGS.sdf_shard = (0, 1) if parallel else None
    torch.manual_seed(1)
    sc._zero_grad()
    r = sc.geometry.tick_init(sc.glctx, sc.target(bg), None, sc.material, sc.loss_fn, 0, None)
    HOOKED.clear()
    if os.environ.get('HOOKS') == '1':           # gradients of intermediate tensors, cloned on-stream (no host sync)
        d = sc.geometry.last_mesh_dict
        for nm, t in (('sdf', d.get('sdf')), ('verts', d['imesh'].v_pos), ('posed', d['deform_imesh'].v_pos), ('msdf', d.get('msdf'))):
            if torch.is_tensor(t) and t.requires_grad:
                t.register_hook(lambda g_, nm=nm: HOOKED.append((nm, g_.detach().clone())))
    r['d3h_total'].backward()
    if parallel:
        sc.allreduce_grads()
    g = [sc.geometry.deform.grad.clone(), sc.material['kd_ks'].encoder.params.grad.clone()] + [p.grad.clone() for p in sc.geometry.sdf_net.parameters()]
    torch.cuda.synchronize()
    g = g + [t for _, t in sorted(HOOKED, key=lambda kv: kv[0])]
    return float(r['d3h_total'].detach()), g, {k: float(v) for k, v in r.items() if torch.is_tensor(v) and v.numel() == 1}

# checksums of every autograd.Function backward's inputs / outputs, in execution order (to find the FIRST quantity that differs)
TRACE = []
def _wrap_all():
    import importlib, pkgutil, d3h, geometry, render
    seen = set()
    for pkg in (d3h, geometry, render):
        for mi in pkgutil.iter_modules(pkg.__path__, pkg.__name__ + '.'):
            try:
                mod = importlib.import_module(mi.name)
            except Exception:
                continue
            for nm, cls in list(vars(mod).items()):
                if isinstance(cls, type) and issubclass(cls, torch.autograd.Function) and cls is not torch.autograd.Function and cls not in seen and 'backward' in vars(cls):
                    seen.add(cls)
                    f = cls.backward
                    def g(ctx, *a, _f=f, _n=f'{mi.name.split(".")[-1]}.{nm}'):
                        out = _f(ctx, *a)
                        cs = lambda ts: [None if (t is None or not torch.is_tensor(t)) else torch.stack([t.double().sum(), t.double().abs().sum()]) for t in (ts if isinstance(ts, (tuple, list)) else (ts,))]      # device scalars: no host sync inside the backward
                        TRACE.append((_n, cs(a), cs(out)))
                        return out
                    cls.backward = staticmethod(g)
if os.environ.get('TRACE') == '1':
    _wrap_all()
HOOKED = []
modes = os.environ.get('MODES', 'FT')          # F = plain, T = arena
ref = None
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 10):
    for m in modes:
        TRACE.clear()
        l, g, terms = tick(m == 'T')
        tr = [(n_, [None if c is None else tuple(c.tolist()) for c in i_], [None if c is None else tuple(c.tolist()) for c in o_]) for n_, i_, o_ in TRACE]
        if ref is None:
            ref = (l, g, terms, tr)
            continue
        bad = [(n, float((a - b).norm() / (a.norm() + 1e-20))) for n, a, b in zip(names, ref[1], g) if (a - b).norm() > 1e-4 * a.norm() + 1e-9]
        print(it, m, 'loss diff', abs(l - ref[0]) / abs(ref[0]), 'bad:', [b for b in bad if not b[0].startswith('net.')][:8], 'nets bad', sum(b[0].startswith('net.') for b in bad), flush=True)
        if bad and tr and m == modes[0]:
            def close(a, b):
                if a is None or b is None:
                    return a is b
                return abs(a[0] - b[0]) <= 1e-5 * (abs(a[1]) + 1e-12) and abs(a[1] - b[1]) <= 1e-5 * (abs(a[1]) + 1e-12)
            for (n1, i1, o1), (n2, i2, o2) in zip(ref[3], tr):
                bi = [k for k, (a, b) in enumerate(zip(i1, i2)) if not close(a, b)]
                bo = [k for k, (a, b) in enumerate(zip(o1, o2)) if not close(a, b)]
                if n1 != n2 or bi or bo:
                    print('   first differing backward:', n1, n2, 'inputs', bi, [(i1[k], i2[k]) for k in bi][:2], 'outputs', bo, [(o1[k], o2[k]) for k in bo][:2])
                    break
        if bad and len(g) > len(names) - 4 and os.environ.get('HOOKS') == '1':
            hv_ref, hv = ref[1][-1], g[-1]                    # h:verts
            dd = (hv - hv_ref)
            rows = torch.nonzero(dd.abs().sum(dim=-1) > 1e-6 * hv_ref.abs().max()).reshape(-1)
            print('   h:verts rows differing', rows.numel(), 'of', hv.shape[0], 'first', rows[:8].tolist(), 'last', rows[-4:].tolist())
            for r_ in rows[:4].tolist():
                print('      row', r_, 'ref', [round(v, 6) for v in hv_ref[r_].tolist()], 'now', [round(v, 6) for v in hv[r_].tolist()])
        if bad:
            d = (ref[1][0] - g[0]).abs().sum(dim=-1)
            nz = torch.nonzero(d > 1e-6 * ref[1][0].abs().max()).reshape(-1)
            print('   deform rows differing', nz.numel(), nz[:10].tolist(), 'terms', {k: (terms[k], ref[2][k]) for k in terms if abs(terms[k] - ref[2][k]) > 1e-6 * abs(ref[2][k]) + 1e-12})
