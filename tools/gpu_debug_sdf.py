import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'd3human-code_amd'))
import torch, torch.nn as nn, numpy as np
from d3h import _lib as L
torch.manual_seed(0)
dims = [(39, 256), (256, 256), (256, 256), (256, 256), (295, 256), (256, 256), (256, 256), (256, 1)]
sd = {}
for li, (i, o) in enumerate(dims):
    l = nn.Linear(i, o); sd[f'net.{2*li}.weight'] = l.weight.detach(); sd[f'net.{2*li}.bias'] = l.bias.detach()
c = lambda t: np.ascontiguousarray(t.numpy(), dtype=np.float32)
w0, b0 = c(sd['net.0.weight']), c(sd['net.0.bias'])
wh = np.stack([c(sd[f'net.{i}.weight']) for i in (2, 4, 6, 10, 12)]); bh = np.stack([c(sd[f'net.{i}.bias']) for i in (2, 4, 6, 10, 12)])
w4, b4 = c(sd['net.8.weight']), c(sd['net.8.bias']); w7, b7 = c(sd['net.14.weight']), c(sd['net.14.bias'])
n = 64
x = (torch.rand(n, 3) * 2.4 - 1.2); xs = c(x)
# emulator
E = ctypes.CDLL(os.path.join(ROOT, 'tests/emul/libd3h_emul.so'))
E.d3h_sdf_mlp_wpack_floats.restype = ctypes.c_int64; E.d3h_sdf_mlp_act_floats.restype = ctypes.c_int64; E.d3h_sdf_mlp_act_floats.argtypes = [ctypes.c_int64]
P = lambda a: a.ctypes.data_as(ctypes.c_void_p)
wp_e = np.zeros(E.d3h_sdf_mlp_wpack_floats(), np.float32)
E.d3h_sdf_mlp_pack(P(w0), P(b0), P(wh), P(bh), P(w4), P(b4), P(w7), P(b7), P(wp_e), None)
out_e = np.zeros(n, np.float32); act_e = np.zeros(E.d3h_sdf_mlp_act_floats(n), np.float32)
E.d3h_sdf_mlp_fwd(P(xs), None, ctypes.c_float(0), P(wp_e), P(out_e), None, P(act_e), ctypes.c_int64(n), None)
# gpu
lib = L.lib(); dev = 'cuda'
T = lambda a: torch.from_numpy(a).to(dev).contiguous()
wp = torch.zeros(lib.d3h_sdf_mlp_wpack_floats(), device=dev)
keep=[T(a) for a in (w0,b0,wh,bh,w4,b4,w7,b7)]
L.check(lib.d3h_sdf_mlp_pack(*[L.ptr(t) for t in keep], L.ptr(wp), L.stream()), 'pack')
torch.cuda.synchronize()
print('wpack diff', np.abs(wp.cpu().numpy() - wp_e).max())
out = torch.zeros(n, device=dev); act = torch.zeros(lib.d3h_sdf_mlp_act_floats(n), device=dev)
xd=T(xs)
L.check(lib.d3h_sdf_mlp_fwd(L.ptr(xd), None, L.f32(0), L.ptr(wp), L.ptr(out), None, L.ptr(act), L.i64(n), L.stream()), 'fwd')
torch.cuda.synchronize()
a = act.cpu().numpy().reshape(-1, 7, 8, 4, 64, 4); ae = act_e.reshape(-1, 7, 8, 4, 64, 4)
for l in range(7):
    d = np.abs(a[:, l] - ae[:, l])
    print('layer', l, 'max diff', d.max(), 'emul absmax', np.abs(ae[:, l]).max(), 'argmax', np.unravel_index(d.argmax(), d.shape))
    if l == 0:
        for rb in range(8):
            print('  rb', rb, np.abs(a[0, l, rb] - ae[0, l, rb]).max())
print('out diff', np.abs(out.cpu().numpy() - out_e).max())
print(a[0, 0, 0, 0, :4], ae[0, 0, 0, 0, :4])
