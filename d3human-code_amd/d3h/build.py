"""Build libd3h_hip.so (all csrc/*.hip) for gfx950 with hipcc, in-tree.

`python d3human-code_amd/d3h/build.py` or `__graft_entry__.build()`.  hipcc cross-compiles without a GPU.
"""
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(os.path.dirname(HERE), 'csrc')
OUT = os.path.join(HERE, 'libd3h_hip.so')
FLAGS = ['--offload-arch=gfx950', '-O3', '-ffp-contract=off', '-fPIC', '-shared', '-std=c++17',
         '-Wno-unused-value', '-Wno-pass-failed']


def needs_build():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    srcs = glob.glob(os.path.join(CSRC, '*.hip')) + glob.glob(os.path.join(CSRC, '*.h'))
    return any(os.path.getmtime(s) > t for s in srcs)


def build(force=False, verbose=True):
    if not force and not needs_build():
        return OUT
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    srcs = sorted(glob.glob(os.path.join(CSRC, '*.hip')))
    objs = []
    procs = []
    os.makedirs(os.path.join(CSRC, 'obj'), exist_ok=True)
    for s in srcs:
        o = os.path.join(CSRC, 'obj', os.path.basename(s)[:-4] + '.o')
        objs.append(o)
        if force or not os.path.exists(o) or os.path.getmtime(o) < max(
                [os.path.getmtime(s)] + [os.path.getmtime(h) for h in glob.glob(os.path.join(CSRC, '*.h'))]):
            cmd = [hipcc] + [f for f in FLAGS if f != '-shared'] + ['-c', s, '-o', o, '-I', CSRC]
            procs.append((s, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for s, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            sys.stderr.write(out.decode())
            raise RuntimeError(f'hipcc failed on {s}')
    cmd = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', OUT] + objs
    subprocess.check_call(cmd)
    if verbose:
        print(f'[d3h] built {OUT} from {len(srcs)} sources')
    return OUT


if __name__ == '__main__':
    build(force='--force' in sys.argv)
