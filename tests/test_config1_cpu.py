"""BASELINE configs[0] (CPU plumbing: template mesh, mask-only, pure-PyTorch/numpy rasteriser) runs end to end on the oracle."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_config1_cpu_runner_fits_the_translation():
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'run_config1_cpu.py'), '--iters', '4', '--res', '128', '--grid', '16'],
                       capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    d = json.loads(p.stdout.strip().splitlines()[-1])
    assert d['mask_loss_last'] < d['mask_loss_first'] and d['iters_per_s'] > 0 and d['mesh_faces'] > 100
