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
