"""bench.py --gpus N starts its own ranks (north star: one process per GPU over RCCL; the reference has no launcher, train.py:1645-1655).
CPU: the launcher's failure semantics -- a rank that dies gives a non-zero exit and NO result line, a mislabelled launch is refused.
-m gpu: `python bench.py --gpus 2` end to end on a one-GPU box over the gloo / shared-device loopback (RCCL refuses two ranks on one
device; everything else of the N > 1 path runs: rendezvous, parameter broadcast, the gradient bucket, barriers, max-over-ranks timing)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, 'bench.py')


def _run(args, env=None, timeout=600):
    e = dict(os.environ)
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, BENCH] + args, env=e, capture_output=True, text=True, timeout=timeout)


def _has_gpu():
    import torch
    return torch.cuda.is_available()


@pytest.mark.skipif(_has_gpu(), reason='needs a box WITHOUT a GPU: the ranks must die')
def test_self_launched_ranks_that_die_give_nonzero_exit_and_no_line():
    r = _run(['--gpus', '2', '--steps', '2', '--warmup', '1', '--no-cpu-baseline', '--no-extras'])
    assert r.returncode != 0
    assert '{"metric"' not in r.stdout
    assert 'needs an MI355X' in r.stderr or 'exited with code' in r.stderr


def test_mislabelled_launch_is_refused():
    """a launcher that started 1 rank for --gpus 2 (or vice versa) must not produce a line that says n_gpus = 1"""
    r = _run(['--gpus', '2', '--steps', '2', '--warmup', '1'], env={'WORLD_SIZE': '1', 'RANK': '0', 'LOCAL_RANK': '0'})
    assert r.returncode != 0 and '{"metric"' not in r.stdout
    assert 'refusing' in r.stderr


@pytest.mark.gpu
def test_gpu_bench_gpus_2_self_launch_loopback(gpu):
    r = _run(['--gpus', '2', '--steps', '4', '--warmup', '2', '--config', '2', '--prefit', '50', '--no-cpu-baseline', '--no-extras', '--both-modes'],
             env={'D3H_DIST_BACKEND': 'gloo', 'D3H_SHARE_GPU': '1'}, timeout=900)
    assert r.returncode == 0, r.stderr[:3000] + '\n...\n' + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{"metric"')]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d['n_gpus'] == 2 and d['config']['world_size'] == 2 and d['scaling'] == 'weak'
    assert d['config']['collective']['bytes'] > 0 and d['config']['collective']['calls'] == 4
    assert d['value'] > 0
    # the line validates itself: the ranks a collective saw, the mode of the frame-independent work, and the other mode's rate beside it
    assert d['config']['rccl_ranks_seen'] == 2 and d['config']['mode'] == 'shard'
    om = d['config']['other_mode']
    assert om['mode'] == 'replicate' and om['value'] > 0 and om['collectives_per_step'] == 1
