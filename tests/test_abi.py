"""CPU-side checks of the drop-in boundary: the shared object builds, loads and exports
every symbol include/callireader_hip.h declares (no compute calls: there is no GPU here)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, 'include', 'callireader_hip.h')


def declared_symbols():
    src = open(HEADER).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(cr_[a-z0-9_]+)\s*\(', src)))


def test_library_builds_and_exports_header_symbols():
    from callireader_amd import build
    lib_path = build.build()
    lib = ctypes.CDLL(lib_path)
    syms = declared_symbols()
    assert len(syms) >= 20
    for s in syms:
        assert hasattr(lib, s), f'{s} declared in the header but not exported'


def test_binding_covers_header():
    from callireader_amd import _binding as B
    assert sorted(B.SIGNATURES) == declared_symbols()
    assert B.lib.cr_abi_version() == B.ABI_VERSION


def test_error_reporting_without_gpu():
    from callireader_amd import _binding as B
    # argument validation happens before any device call
    assert B.lib.cr_create(0, None, None) == -1
    assert b'null' in B.lib.cr_last_error()
    assert B.lib.cr_op_gemm(0, None, 8, None, 8, None, 8, None, None, None, 0, 4, 4, 7, 0, None) != 0
    assert b'cr_op_gemm' in B.lib.cr_last_error()


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, 'callireader_amd')
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith('.py'):
                txt = open(os.path.join(dp, f)).read()
                assert not re.search(r'^\s*(from|import)\s+oracle\b', txt, flags=re.M), f


def test_diagnostic_builds_are_quarantined(tmp_path):
    """Round-4 verdict, item 8: the knock-out / poison / stamp blocks inside the product kernels are reachable only through csrc/diag.hpp.
    (a) the product library reports no diagnostic flags; (b) a diagnostic macro without the variant builder's -DCR_DIAG_BUILD does not compile;
    (c) a -DCR_KO_EPI library (scripts/build_variant.py: wrong results by design) reports its flag, carries another build id than the product, and
    placed at the product's default path it FAILS TO LOAD with a clear message; named explicitly in CR_HIP_LIB it loads (the A/B route)."""
    import shutil
    import subprocess
    import sys
    from callireader_amd import build, _binding as B
    assert B.lib.cr_build_flags() == b''
    csrc = os.path.join(ROOT, 'callireader_amd', 'csrc')
    r = subprocess.run(['hipcc', '--offload-arch=gfx950', '-std=c++17', '-DCR_KO_XFRAG=1', '-E', os.path.join(csrc, 'gemm_skinny.hip'), '-o', os.devnull],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    assert r.returncode != 0 and b'build_variant.py' in r.stdout
    subprocess.check_call([sys.executable, os.path.join(ROOT, 'scripts', 'build_variant.py'), 'ko_quarantine_test', 'gemm_skinny.hip', '-DCR_KO_XFRAG=1'],
                          stdout=subprocess.DEVNULL)
    variant = os.path.join(ROOT, 'ab', 'libko_quarantine_test.so')
    blob = open(variant, 'rb').read()
    i = blob.find(b'CR_BUILD_ID=')
    assert i >= 0 and blob[i + 12:i + 28].decode() != build.source_hash()
    # a copy of the package whose DEFAULT library is the variant
    pkg = tmp_path / 'callireader_amd'
    (pkg / 'csrc').mkdir(parents=True)
    for f in os.listdir(os.path.join(ROOT, 'callireader_amd')):
        if f.endswith('.py'):
            shutil.copy(os.path.join(ROOT, 'callireader_amd', f), pkg / f)
    shutil.copy(variant, pkg / 'csrc' / 'libcallireader_hip.so')
    env = {k: v for k, v in os.environ.items() if k != 'CR_HIP_LIB'}
    code = 'import sys; sys.path.insert(0, sys.argv[1]); import callireader_amd._binding as B; print(B.lib.cr_build_flags().decode())'
    r = subprocess.run([sys.executable, '-c', code, str(tmp_path)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, cwd=str(tmp_path))
    assert r.returncode != 0 and b'DIAGNOSTIC build (CR_KO_XFRAG)' in r.stderr, r.stderr[-400:]
    r = subprocess.run([sys.executable, '-c', code, str(tmp_path)], env=dict(env, CR_HIP_LIB=variant), stdout=subprocess.PIPE, stderr=subprocess.PIPE, cwd=str(tmp_path))
    assert r.returncode == 0 and r.stdout.strip() == b'CR_KO_XFRAG' and b'diagnostic library' in r.stderr, r.stderr[-400:]
