"""ctypes binding of libcallireader_hip.so (include/callireader_hip.h).

This is the stub a maintainer of the reference would add next to
InternVL/modeling_internvl_chat.py.  There is no fallback:
if the shared object is missing or does not export the ABI, import fails loudly.
"""
import ctypes as C
import os

import torch  # noqa: F401  -- FIRST: PyTorch-ROCm bundles its own libamdhip64; loading ours before it would start a
#                              second HIP runtime in the process ("no ROCm-capable device is detected")

_LIB_PATH = os.environ.get('CR_HIP_LIB') or os.path.join(os.path.dirname(os.path.abspath(__file__)), 'csrc', 'libcallireader_hip.so')   # CR_HIP_LIB: A/B a second build (development aid)

CR_OK = 0
CR_BF16, CR_F32, CR_I64, CR_I32 = 0, 1, 2, 3
ABI_VERSION = 10


class ModelDesc(C.Structure):
    _fields_ = [('vit_layers', C.c_int32), ('rs_depth', C.c_int32), ('llm_layers', C.c_int32),
                ('vocab', C.c_int32), ('max_pos', C.c_int32), ('vit_ln_eps', C.c_float),
                ('rms_eps', C.c_float), ('reserved', C.c_int32 * 8)]


vp, i32, i64, f32 = C.c_void_p, C.c_int, C.c_int64, C.c_float

# name -> (restype, argtypes): exactly the prototypes of include/callireader_hip.h
SIGNATURES = {
    'cr_last_error': (C.c_char_p, []),
    'cr_abi_version': (i32, []),
    'cr_build_id': (C.c_char_p, []),
    'cr_build_flags': (C.c_char_p, []),
    'cr_diag_register': (i32, [C.c_char_p]),
    'cr_create': (i32, [i32, C.POINTER(ModelDesc), C.POINTER(vp)]),
    'cr_destroy': (i32, [vp]),
    'cr_load_weight': (i32, [vp, C.c_char_p, vp, i32, C.POINTER(i64), i32, i32, vp]),
    'cr_finalize': (i32, [vp, vp]),
    'cr_vit_forward': (i32, [vp, vp, i32, vp, vp]),
    'cr_project': (i32, [vp, vp, i32, vp, vp]),
    'cr_extract_feature': (i32, [vp, vp, i32, vp, vp]),
    'cr_preprocess': (i32, [vp, vp, i32, i32, vp, i32, vp, vp, i32, vp]),
    'cr_resample': (i32, [vp, vp, i32, vp, vp]),
    'cr_vq': (i32, [vp, vp, i32, vp, vp, vp]),
    'cr_orderformer': (i32, [vp, vp, i32, i32, vp, vp]),
    'cr_denorm': (i32, [vp, vp, vp, vp, i32, i32, vp, vp, vp]),
    'cr_embed_splice': (i32, [vp, vp, i32, vp, i32, i64, vp, i32, i64, vp, vp]),
    'cr_kv_alloc': (i32, [vp, i32, i32, C.POINTER(vp)]),
    'cr_kv_free': (i32, [vp]),
    'cr_kv_length': (i32, [vp, i32]),
    'cr_kv_reset': (i32, [vp, i32, vp]),
    'cr_kv_read': (i32, [vp, i32, i32, i32, i32, vp, vp]),
    'cr_kv_generated': (i32, [vp, i32, C.POINTER(i64), i32, vp]),
    'cr_llm_prefill': (i32, [vp, vp, i32, vp, i32, f32, vp, vp]),
    'cr_llm_prefill_batch': (i32, [vp, vp, C.POINTER(C.c_int32), i32, vp, C.POINTER(C.c_int32), f32, vp, vp]),
    'cr_llm_hidden_probe': (i32, [vp, vp, i32, i32]),
    'cr_llm_decode': (i32, [vp, vp, C.POINTER(C.c_int32), i32, vp, f32, vp, vp]),
    'cr_share_weights': (i32, [vp, vp]),
    'cr_enable_fp8_decode': (i32, [vp, i32, vp]),
    'cr_enable_fp8_mfma': (i32, [vp, i32, vp]),
    'cr_op_quantize_fp8': (i32, [vp, i64, i32, i32, vp, vp, vp]),
    'cr_profile': (i32, [vp, i32]),
    'cr_profile_read': (i32, [vp, C.POINTER(C.c_double)]),
    'cr_profile_stats': (i32, [vp, C.POINTER(i64)]),
    'cr_op_gemm': (i32, [i32, vp, i64, vp, i64, vp, i64, vp, vp, vp, i64, i32, i32, i32, i32, vp]),
    'cr_op_layernorm': (i32, [vp, vp, vp, vp, i64, i32, f32, i32, vp]),
    'cr_op_rmsnorm': (i32, [vp, vp, vp, i64, i32, f32, vp]),
    'cr_op_norm_fp8': (i32, [vp, vp, vp, i64, i32, f32, vp, vp, vp, vp, vp]),
    'cr_op_gemm_q8': (i32, [vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, vp]),
    'cr_op_decode_swizzle': (i32, [i32, vp, i64, i32, i32, vp, vp]),
    'cr_op_decode_gemm': (i32, [i32, i32, vp, i64, i32, i32, i32, vp, i64, vp, vp, f32, vp, vp, i64, vp, vp, vp, vp, vp, vp, vp, i32, vp]),
    'cr_op_attention': (i32, [vp, vp, vp, vp, C.POINTER(i64), i32, i32, i32, i32, i32, i32, i32, i32, f32, f32, vp]),
    'cr_op_decode_attention_scratch_floats': (i64, [i32, i32]),
    'cr_op_decode_attention': (i32, [i32, vp, vp, vp, i32, vp, vp, i32, i32, f32, vp, vp, vp]),
}


class PrepJob(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ('sx0', 'sy0', 'sw', 'sh', 'ow', 'oh', 'mode', 'tile0', 'cols', 'left', 'top')]


class CalliReaderError(RuntimeError):
    pass


def _load():
    if not os.path.exists(_LIB_PATH):
        raise ImportError(f'{_LIB_PATH} is missing: run `python -c "import __graft_entry__ as g; g.build()"` '
                          '(there is no CPU or PyTorch fallback for the hot path)')
    lib = C.CDLL(_LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the ABI is incomplete
        fn.restype = res
        fn.argtypes = args
    if lib.cr_abi_version() != ABI_VERSION:
        raise ImportError(f'ABI version mismatch: library {lib.cr_abi_version()} vs binding {ABI_VERSION}')
    # diagnostic builds (csrc/diag.hpp: knock-outs, poison, stamps -- several give WRONG RESULTS by design) never load by accident: only when CR_HIP_LIB
    # names that very file
    flags = (lib.cr_build_flags() or b'').decode().split()
    if flags:
        if not os.environ.get('CR_HIP_LIB'):
            raise ImportError(f'{_LIB_PATH} is a DIAGNOSTIC build ({" ".join(flags)}): it must not sit at the product\'s path -- rebuild with '
                              '`python -c "import __graft_entry__ as g; g.build()"`; a variant library is loaded by naming it in CR_HIP_LIB')
        import sys
        print(f'[callireader_amd] diagnostic library {_LIB_PATH}: {" ".join(flags)}', file=sys.stderr)
    return lib


lib = _load()


def check(code, what=''):
    if code != CR_OK:
        msg = lib.cr_last_error()
        raise CalliReaderError(f'{what}: error {code}: {msg.decode() if msg else ""}')
