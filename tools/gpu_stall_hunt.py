"""what is the host doing during the one-off ~80 ms iteration (it ~ 25-35 of a fresh process)?  A sampling thread records the main
thread's Python stack every 2 ms; the stacks sampled inside the slowest iteration are printed.   python tools/gpu_stall_hunt.py"""
import os, sys, time, threading, traceback, collections
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), 'd3human-code_amd'))
import torch
torch.cuda.set_device(0)
from d3h import scene
sc = scene.Scene(device='cuda:0', prefit_steps=300, visualize_watertight=True, res=1024, grid_n=63, n_frames=4, loss_set='full')
main_id = threading.main_thread().ident
samples, stop = [], False
def sampler():
    while not stop:
        f = sys._current_frames().get(main_id)
        if f is not None:
            samples.append((time.time(), ''.join(traceback.format_stack(f, limit=12))))
        time.sleep(0.002)
th = threading.Thread(target=sampler, daemon=True); th.start()
import gc
gcev = []
def cb(phase, info):
    if phase == 'start':
        gcev.append([time.time(), info['generation'], None, None])
    else:
        gcev[-1][2] = time.time() - gcev[-1][0]; gcev[-1][3] = info.get('collected')
gc.callbacks.append(cb)
spans = []
torch.cuda.synchronize()
for i in range(60):
    t0 = time.time()
    sc.step()
    spans.append((t0, time.time()))
torch.cuda.synchronize()
stop = True
dur = [(b - a, i) for i, (a, b) in enumerate(spans)]
print('host ms per iteration (no sync):', ' '.join(f'{d * 1e3:.1f}' for d, _ in dur))
d, i = max(dur)
print(f'slowest iteration {i}: {d * 1e3:.1f} ms host time')
a, b = spans[i]
print('gc runs inside it:', [(g, f'{(d or 0) * 1e3:.1f} ms', col) for t, g, d, col in gcev if a <= t <= b], '| all gen2:', [(f'{(d or 0) * 1e3:.1f} ms') for t, g, d, col in gcev if g == 2], 'objects tracked', len(gc.get_objects()))
c = collections.Counter(s for t, s in samples if a <= t <= b)
for s, n in c.most_common(3):
    print(f'--- {n} samples ---\n{s}')
