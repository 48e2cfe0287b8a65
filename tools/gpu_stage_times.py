"""dev tool: per-stage wall times of one config-3 iteration (with syncs between stages)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'd3human-code_amd'))
import torch
from d3h import scene
sc = scene.Scene(res=1024, grid_n=63, n_frames=4, device='cuda', prefit_steps=300, loss_set='full', visualize_watertight=True)
for _ in range(3): sc.step()
torch.cuda.synchronize()
g = sc.geometry
def T(fn, n=5):
    torch.cuda.synchronize(); t=time.time()
    for _ in range(n): r = fn()
    torch.cuda.synchronize(); return (time.time()-t)/n*1e3, r
bg = torch.rand(4,1024,1024,3,device='cuda'); tgt = sc.target(bg)
t, _ = T(lambda: g._sdf_sweep()); print(f'sdf sweep fwd        {t:7.2f} ms')
vd, sdf = g._sdf_sweep()
t, o = T(lambda: g.gshell_tets(vd, sdf, g.msdf, g.indices)); print(f'marching tets fwd    {t:7.2f} ms')
t, d = T(lambda: g.getMesh_init(sc.material, target=tgt)); print(f'getMesh_init (all)   {t:7.2f} ms')
t, d = T(lambda: g.render_init(sc.glctx, tgt, None, sc.material, buffers=sc.FLAGS.render_buffers)); print(f'render_init (all fwd){t:7.2f} ms')
t, r = T(lambda: g.tick_init(sc.glctx, tgt, None, sc.material, sc.loss_fn, 10, None)); print(f'tick_init fwd        {t:7.2f} ms')
def fb():
    sc._zero_grad()
    r = g.tick_init(sc.glctx, tgt, None, sc.material, sc.loss_fn, 10, None)
    (r['reg_loss']+r['normal_loss']+r['msk_loss']+r['ssim_loss']).backward()
t, _ = T(fb); print(f'tick fwd+bwd         {t:7.2f} ms')
def eik():
    return g._eikonal(g.last_mesh_dict['sampled_pts'], 10)
t, e = T(eik); print(f'eikonal fwd          {t:7.2f} ms')
def eikb():
    e = g._eikonal(g.last_mesh_dict['sampled_pts'], 10); e.backward()
t, _ = T(eikb); print(f'eikonal fwd+bwd      {t:7.2f} ms')
def opt():
    sc._optimizer_step()
t, _ = T(opt); print(f'adam steps           {t:7.2f} ms')
t, _ = T(sc.step); print(f'full step            {t:7.2f} ms')
