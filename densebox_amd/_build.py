"""Build libdensebox_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'csrc')
LIB = os.path.join(CSRC, 'libdensebox_hip.so')
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-Wall', '-Wno-unused-function',
         '-Wno-unused-variable', '-Wno-unused-but-set-variable']


def _sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith('.hip'))


def _newer(src, dst):
    if not os.path.exists(dst):
        return True
    deps = [src] + [os.path.join(CSRC, h) for h in os.listdir(CSRC) if h.endswith('.hpp')]
    deps.append(os.path.join(os.path.dirname(os.path.dirname(CSRC)), 'include', 'densebox_hip.h'))
    return any(os.path.getmtime(d) > os.path.getmtime(dst) for d in deps if os.path.exists(d))


def build(force=False, verbose=True):
    srcs = _sources()
    objs = []
    jobs = []
    for s in srcs:
        src = os.path.join(CSRC, s)
        obj = os.path.join(CSRC, s[:-4] + '.o')
        objs.append(obj)
        if force or _newer(src, obj):
            jobs.append((src, obj))

    def cc(job):
        src, obj = job
        cmd = [HIPCC] + FLAGS + ['-c', src, '-o', obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        return src, r.returncode, r.stdout + r.stderr
    if jobs:
        with ThreadPoolExecutor(max_workers=min(8, len(jobs))) as ex:
            for src, rc, out in ex.map(cc, jobs):
                if verbose and out.strip():
                    print(out, file=sys.stderr)
                if rc != 0:
                    raise RuntimeError('hipcc failed on %s:\n%s' % (src, out))
    stale = not os.path.exists(LIB) or any(os.path.getmtime(o) > os.path.getmtime(LIB) for o in objs)
    if jobs or stale:
        cmd = [HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError('link failed:\n' + r.stdout + r.stderr)
    return LIB


if __name__ == '__main__':
    print(build(force='--force' in sys.argv))
