#!/usr/bin/env python3
"""Development aid: build ab/lib<name>.so = the library with ONE source (or several, comma-separated) recompiled under extra flags (A/B runs
select it with CR_HIP_LIB=ab/lib<name>.so).  usage: build_variant.py <name> <source.hip[,other.hip]> [-DFOO=1 ...]"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from callireader_amd import build as B  # noqa: E402

import hashlib  # noqa: E402

name, src = sys.argv[1], sys.argv[2]
flags = sys.argv[3:]
B.build()
os.makedirs(os.path.join(ROOT, 'ab'), exist_ok=True)
srcs = src.split(',')
# the only door to the diagnostic macros (csrc/diag.hpp): -DCR_DIAG_BUILD; and the variant's cr_build_id() = hash(sources, recompiled files, flags), never the product's
vid = hashlib.sha256((B.source_hash() + ' ' + src + ' ' + ' '.join(flags)).encode()).hexdigest()[:16]
if 'api.hip' not in srcs:
    srcs_id = srcs + ['api.hip']
else:
    srcs_id = srcs
new_objs = []
for one in srcs_id:
    obj = os.path.join(ROOT, 'ab', f'{name}_{one[:-4]}.o')
    own = flags if one in srcs else []
    cmd = ['hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-Wno-unused-value', '-DCR_DIAG_BUILD=1', f'-DCR_BUILD_ID="CR_BUILD_ID={vid}"'] \
        + B.EXTRA_FLAGS.get(one, []) + own + ['-c', os.path.join(B.CSRC, one), '-o', obj]
    subprocess.check_call(cmd)
    new_objs.append(obj)
srcs = srcs_id
objs = [os.path.join(B.CSRC, 'build', os.path.basename(s)[:-4] + '.o') for s in B.sources() if os.path.basename(s) not in srcs] + new_objs
out = os.path.join(ROOT, 'ab', f'lib{name}.so')
subprocess.check_call(['hipcc', '--offload-arch=gfx950', '-shared', '-fPIC', '-o', out] + objs)
print(out)
