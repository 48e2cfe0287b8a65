"""Where do the zero fills of one init-stage iteration come from?  Runs the step on the host emulator (tests/emul) with torch's
zero-filling constructors wrapped, prints call sites by count.  python tools/dbg/fill_sites.py"""
import collections, os, sys, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'd3human-code_amd'))
import torch
from d3h import _lib as L
L._use_emulator_for_tests(os.path.join(ROOT, 'tests', 'emul', 'libd3h_emul.so'))
from d3h.scene import Scene
ell = lambda x: (((x - torch.tensor([0.0, -0.4, 0.0])) / torch.tensor([0.55, 0.8, 0.45])).norm(dim=-1) - 1.0) * 0.4
sc = Scene(res=24, grid_n=4, n_frames=2, device='cpu', prefit_steps=60, loss_set='full', body_verts=300, sdf_fn=ell,
           flags_hook=lambda F: (setattr(F, 'prefit_with_library_path', True), setattr(F, 'eikonal_samples', 128)))
sc.step()
sites = collections.Counter()
def wrap(mod, name):
    f = getattr(mod, name)
    def g(*a, **k):
        fr = [x for x in traceback.extract_stack()[:-1] if 'd3human-code_amd' in x.filename]
        key = name + ' <- ' + ' <- '.join(f'{os.path.basename(x.filename)}:{x.lineno}' for x in fr[-2:][::-1])
        sites[key] += 1
        return f(*a, **k)
    setattr(mod, name, g)
for n in ('zeros', 'zeros_like', 'full', 'ones', 'ones_like', 'full_like', 'cat', 'stack'):
    wrap(torch, n)
zt = torch.Tensor.zero_
def z(self):
    fr = [x for x in traceback.extract_stack()[:-1] if 'd3human-code_amd' in x.filename]
    sites['zero_ <- ' + ' <- '.join(f'{os.path.basename(x.filename)}:{x.lineno}' for x in fr[-2:][::-1])] += 1
    return zt(self)
torch.Tensor.zero_ = z
nz = torch.Tensor.new_zeros
def nzw(self, *a, **k):
    fr = [x for x in traceback.extract_stack()[:-1] if 'd3human-code_amd' in x.filename]
    sites['new_zeros <- ' + ' <- '.join(f'{os.path.basename(x.filename)}:{x.lineno}' for x in fr[-2:][::-1])] += 1
    return nz(self, *a, **k)
torch.Tensor.new_zeros = nzw
sc.step()
for k, v in sites.most_common():
    print(v, k)
