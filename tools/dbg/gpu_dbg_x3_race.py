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
sc = Scene(res=128, grid_n=12, n_frames=2, device='cuda', prefit_steps=150, loss_set='full', body_verts=2048, sdf_fn=ell)
bg = torch.rand(2, 128, 128, 3, device='cuda')
names = ['deform', 'table'] + [n for n, _ in sc.geometry.sdf_net.named_parameters()]

def tick(parallel):
    sc.world = 2 if parallel else 1
    sc.FLAGS.sdf_shard = (0, 1) if parallel else None
    torch.manual_seed(1)
    sc._zero_grad()
    r = sc.geometry.tick_init(sc.glctx, sc.target(bg), None, sc.material, sc.loss_fn, 0, None)
    r['d3h_total'].backward()
    if parallel:
        sc.allreduce_grads()
    g = [sc.geometry.deform.grad.clone(), sc.material['kd_ks'].encoder.params.grad.clone()] + [p.grad.clone() for p in sc.geometry.sdf_net.parameters()]
    torch.cuda.synchronize()
    return float(r['d3h_total'].detach()), g, {k: float(v) for k, v in r.items() if torch.is_tensor(v) and v.numel() == 1}

modes = os.environ.get('MODES', 'FT')          # F = plain, T = arena
ref = None
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 10):
    for m in modes:
        l, g, terms = tick(m == 'T')
        if ref is None:
            ref = (l, g, terms)
            continue
        bad = [(n, float((a - b).norm() / (a.norm() + 1e-20))) for n, a, b in zip(names, ref[1], g) if (a - b).norm() > 1e-4 * a.norm() + 1e-9]
        print(it, m, 'loss diff', abs(l - ref[0]) / abs(ref[0]), 'bad:', bad[:6], flush=True)
        if bad:
            d = (ref[1][0] - g[0]).abs().sum(dim=-1)
            nz = torch.nonzero(d > 1e-6 * ref[1][0].abs().max()).reshape(-1)
            print('   deform rows differing', nz.numel(), nz[:10].tolist(), 'terms', {k: (terms[k], ref[2][k]) for k in terms if abs(terms[k] - ref[2][k]) > 1e-6 * abs(ref[2][k]) + 1e-12})
