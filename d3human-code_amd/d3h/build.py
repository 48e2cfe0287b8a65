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


def _deps(path, seen=None):
    """`path` and every file of csrc/ it includes, recursively (deform_mlp*.hip ARE sdf_mlp*.hip compiled with other macros: an object that only
    looked at its own source's time stamp stayed stale when the included source changed)"""
    import re
    seen = set() if seen is None else seen
    if path in seen or not os.path.exists(path):
        return seen
    seen.add(path)
    for m in re.finditer(r'^\s*#\s*include\s+"([^"]+)"', open(path).read(), re.M):
        _deps(os.path.join(CSRC, m.group(1)), seen)
    return seen


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
                [os.path.getmtime(d) for d in _deps(s)] + [os.path.getmtime(h) for h in glob.glob(os.path.join(CSRC, '*.h'))]):
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
