"""Build libcallireader_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

The library carries the hash of the sources it was compiled from (`cr_build_id()`, include/callireader_hip.h): a build is
stale when that hash differs from the sources on disk -- file times do not survive a copy to another box, contents do."""
import hashlib
import os
import shutil
import subprocess
import sys

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'csrc')
LIB = os.path.join(CSRC, 'libcallireader_hip.so')
HEADER = os.path.join(os.path.dirname(os.path.dirname(CSRC)), 'include', 'callireader_hip.h')


# attention.hip: scores never hold a NaN (masked keys are -inf, and an all-masked row is never exponentiated), and
# without the assumption every fmaxf on an accumulator costs an extra canonicalising v_max_f32
# attention_vit.hip, -fno-slp-vectorize: the SLP pass packs the row-sum adds into v_pk_add_f32, which issue slower beside MFMAs than the
# plain adds they replace (guide: 'packed f32 VALU ... an anti-lever beside MFMAs'): 1.453 -> 1.432 ms per 255-tile launch, same bits
EXTRA_FLAGS = {'attention.hip': ['-fno-honor-nans', '-fno-slp-vectorize'], 'attention_vit.hip': ['-fno-honor-nans', '-fno-slp-vectorize']}      # (attention_decode.hip compares with -inf on purpose: default flags)


# the product build takes no -D flags: the diagnostic macros of csrc/diag.hpp are reachable through scripts/build_variant.py only
assert not any(f.startswith('-D') for fl in EXTRA_FLAGS.values() for f in fl), 'callireader_amd/build.py: no -D flags in the product build'


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.hip'))


def source_hash():
    """First 16 hex digits of sha256 over every .hip / .hpp / .inc under csrc/ and the ABI header (names and contents, sorted)."""
    h = hashlib.sha256()
    deps = sources() + sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(('.hpp', '.inc'))) + [HEADER]
    for d in deps:
        h.update(os.path.basename(d).encode() + b'\0')
        with open(d, 'rb') as f:
            h.update(f.read())
        h.update(b'\0')
    for k in sorted(EXTRA_FLAGS):
        h.update((k + ' ' + ' '.join(EXTRA_FLAGS[k])).encode())
    return h.hexdigest()[:16]


def built_id():
    """`cr_build_id()` of the shared object on disk, read from the file (no HIP runtime is started), or None."""
    if not os.path.exists(LIB):
        return None
    with open(LIB, 'rb') as f:
        blob = f.read()
    i = blob.find(b'CR_BUILD_ID=')
    return blob[i + 12:i + 28].decode('ascii', 'replace') if i >= 0 else None


def _stale():
    return built_id() != source_hash()


def build(force=False, verbose=True):
    """Compile every .hip under csrc/ into one shared object.  Returns the path."""
    if not force and not _stale():
        return LIB
    hipcc = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    if not os.path.exists(hipcc):
        raise RuntimeError('hipcc not found: cannot build libcallireader_hip.so')
    objs = []
    procs = []
    bid = source_hash()
    os.makedirs(os.path.join(CSRC, 'build'), exist_ok=True)
    for s in sources():
        o = os.path.join(CSRC, 'build', os.path.basename(s)[:-4] + '.o')
        objs.append(o)
        cmd = [hipcc, '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-Wno-unused-value', f'-DCR_BUILD_ID="CR_BUILD_ID={bid}"'] \
            + EXTRA_FLAGS.get(os.path.basename(s), []) + ['-c', s, '-o', o]
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
    assert built_id() == bid, (built_id(), bid)
    return LIB


if __name__ == '__main__':
    print(build(force='--force' in sys.argv), built_id())
