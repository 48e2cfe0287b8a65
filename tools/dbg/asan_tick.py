"""the f3c shape in small on the ASan + UBSan build of the host emulation (tests/emul/build_emul_asan.sh): sizes that are
multiples of NO tile (135 x 135 pixels, 343 / 729 grid vertices, 1 000 / 777 eikonal samples), one frame, the init-stage stack with the
MobileNetV2-feature normal loss; tick_init + backward + the optimiser step, twice; then the split stage and the seq stage likewise.
Every heap block torch hands to a kernel and every emulated __shared__ array carries red zones: an out-of-bounds access aborts.
    bash tools/dbg/asan_tick.sh"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'd3human-code_amd')]
import torch
from d3h import _lib as L
L._use_emulator_for_tests(os.path.join(ROOT, 'tests', 'emul', 'libd3h_emul_asan.so'))
from d3h.scene import Scene

ell = lambda x: (((x - torch.tensor([0.0, -0.4, 0.0])) / torch.tensor([0.55, 0.8, 0.45])).norm(dim=-1) - 1.0) * 0.4
which = sys.argv[1:] or ['init', 'full', 'split', 'seq']
for ls, res, grid, eik in (('init', 135, 6, 1000), ('full', 67, 8, 777), ('split', 45, 6, 333), ('seq', 45, 6, 0)):
    if ls not in which:
        continue
    torch.manual_seed(0)
    hook = lambda F, ls=ls, eik=eik: (setattr(F, 'prefit_with_library_path', True), setattr(F, 'eikonal_samples', eik),
                                     setattr(F, 'use_perceptual_normal_loss', ls == 'init'))
    sc = Scene(res=res, grid_n=grid, n_frames=1, device='cpu', prefit_steps=120, loss_set=ls, body_verts=301, sdf_fn=ell, flags_hook=hook)
    step = {'split': sc.step_split, 'seq': sc.step_seq}.get(ls, sc.step)
    for i in range(2):
        r = step()
        assert all(torch.isfinite(v).all() for v in r.values() if torch.is_tensor(v)), r
    print(f'asan tick {ls}: res {res}, grid {grid}, eikonal samples {eik}: 2 steps, total {float(r["total"]):.5f} -- no sanitizer report', flush=True)
