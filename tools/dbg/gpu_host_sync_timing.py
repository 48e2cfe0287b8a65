"""Where does the host wait, and where does the GPU wait for the host?  Config-3 step with every device->host read-back (Tensor.tolist on a
device tensor: the two size read-backs of the marching-tets extraction) timed on the host clock, no profiler attached.
A read-back that returns at once = the GPU was already idle when the host asked (host-bound stretch before it); one that blocks = the host was ahead.
gpurun -- 'python tools/dbg/gpu_host_sync_timing.py'"""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'd3human-code_amd'))
import torch
from d3h import scene

sc = scene.Scene(device='cuda', prefit_steps=300, res=512, grid_n=32, n_frames=1, loss_set='mask') if os.environ.get('CONFIG') == '2' else \
    scene.Scene(device='cuda', prefit_steps=300, res=1024, grid_n=63, n_frames=4, loss_set='full')
for _ in range(20):
    sc.step()
torch.cuda.synchronize()
log = []
orig = torch.Tensor.tolist


def timed(self):
    if self.is_cuda:
        t0 = time.perf_counter()
        r = orig(self)
        log.append((t0, time.perf_counter()))
        return r
    return orig(self)


torch.Tensor.tolist = timed
from d3h import mtets as _mt
_w = _mt.TetGrid._wait


def _tw(self, *a):
    t0 = time.perf_counter()
    r = _w(self, *a)
    log.append((t0, time.perf_counter()))
    return r


_mt.TetGrid._wait = _tw
N = 200
marks = []
t_begin = time.perf_counter()
for _ in range(N):
    marks.append(time.perf_counter())
    sc.step()
torch.cuda.synchronize()
t_end = time.perf_counter()
torch.Tensor.tolist = orig
per = (t_end - t_begin) / N * 1e3
print(f'{N} steps: {per:.3f} ms / step; read-backs per step: {len(log) / N:.2f}')
k = round(len(log) / N)
import statistics as st
for j in range(k):
    w = [(b - a) * 1e6 for i, (a, b) in enumerate(log) if i % k == j]
    print(f'read-back #{j + 1}: blocked median {st.median(w):.0f} us, p10 {sorted(w)[len(w) // 10]:.0f}, p90 {sorted(w)[9 * len(w) // 10]:.0f}')
# host time from step entry to read-back #1 (prologue + sweep launch), between the read-backs, and from the last read-back to the step's return
ent = marks + [t_end]
pro = [(log[i * k][0] - ent[i]) * 1e6 for i in range(N)]
mid = [(log[i * k + k - 1][0] - log[i * k][1]) * 1e6 for i in range(N)] if k > 1 else [0.0]
tail = [(ent[i + 1] - log[i * k + k - 1][1]) * 1e6 for i in range(N - 1)]
print(f'host: entry -> read-back #1 {st.median(pro):.0f} us; between read-backs {st.median(mid):.0f} us; last read-back -> step return {st.median(tail):.0f} us')
print('(host busy per step, outside the blocked time):', round(st.median(pro) + st.median(mid) + st.median(tail)), 'us')
