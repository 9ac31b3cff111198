"""Build libcallireader_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
import os
import shutil
import subprocess
import sys

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'csrc')
LIB = os.path.join(CSRC, 'libcallireader_hip.so')


# attention.hip: scores never hold a NaN (masked keys are -inf, and an all-masked row is never exponentiated), and
# without the assumption every fmaxf on an accumulator costs an extra canonicalising v_max_f32
EXTRA_FLAGS = {'attention.hip': ['-fno-honor-nans']}


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.hip'))


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = sources() + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.hpp')]
    deps.append(os.path.join(os.path.dirname(os.path.dirname(CSRC)), 'include', 'callireader_hip.h'))
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    """Compile every .hip under csrc/ into one shared object.  Returns the path."""
    if not force and not _stale():
        return LIB
    hipcc = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    if not os.path.exists(hipcc):
        raise RuntimeError('hipcc not found: cannot build libcallireader_hip.so')
    objs = []
    procs = []
    os.makedirs(os.path.join(CSRC, 'build'), exist_ok=True)
    for s in sources():
        o = os.path.join(CSRC, 'build', os.path.basename(s)[:-4] + '.o')
        objs.append(o)
        cmd = [hipcc, '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-Wno-unused-value'] + EXTRA_FLAGS.get(os.path.basename(s), []) + ['-c', s, '-o', o]
        procs.append((cmd, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for cmd, p in procs:
        out = p.communicate()[0].decode()
        if p.returncode != 0:
            raise RuntimeError('hipcc failed: ' + ' '.join(cmd) + '\n' + out)
        if verbose and out.strip():
            print(out, file=sys.stderr)
    link = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + objs
    r = subprocess.run(link, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    if r.returncode != 0:
        raise RuntimeError('link failed:\n' + r.stdout.decode())
    return LIB


if __name__ == '__main__':
    print(build(force='--force' in sys.argv))
