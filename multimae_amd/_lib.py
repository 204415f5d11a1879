"""ctypes binding of libmmae_hip.so (C ABI declared in include/mmae.h).

The prototypes are parsed from the header at import time, so the Python side can
never drift from the ABI, and ``declared_symbols()`` lets the tests check that the
built library exports every declared entry point.

There is NO fallback: if the shared library is missing or a kernel reports an
error, the call raises.  (The product path must fail loudly without the HIP
extension -- the CPU oracle under oracle/ is test infrastructure only.)
"""
from __future__ import annotations

import ctypes
import os
import re
from typing import Dict, List, Tuple

_PKG = os.path.dirname(os.path.abspath(__file__))
HEADER = os.path.join(os.path.dirname(_PKG), 'include', 'mmae.h')
LIB_PATH = os.environ.get('MMAE_LIB') or os.path.join(_PKG, 'libmmae_hip.so')      # MMAE_LIB: an alternative build of the same ABI (A/B experiments)

F32, BF16, F32X3, F32F16 = 0, 1, 2, 3
MXFP8 = 4
F16 = 5            # fp16 storage (the fp32 output adapters' 'h16' mode), see mmae.h
EPI_NONE, EPI_GELU, EPI_DGELU, EPI_GELU_G, EPI_MUL = 0, 1, 2, 3, 4


class GemmDesc(ctypes.Structure):
    """mirror of mmae_gemm_desc"""
    _fields_ = [
        ('A', ctypes.c_void_p), ('B', ctypes.c_void_p), ('C', ctypes.c_void_p),
        ('ab_dtype', ctypes.c_int32), ('c_dtype', ctypes.c_int32),
        ('M', ctypes.c_int32), ('N', ctypes.c_int32), ('K', ctypes.c_int32),
        ('a_trans', ctypes.c_int32), ('b_trans', ctypes.c_int32),
        ('lda', ctypes.c_int64), ('ldb', ctypes.c_int64), ('ldc', ctypes.c_int64),
        ('batch', ctypes.c_int32), ('batch_inner', ctypes.c_int32),
        ('sA_outer', ctypes.c_int64), ('sA_inner', ctypes.c_int64),
        ('sB_outer', ctypes.c_int64), ('sB_inner', ctypes.c_int64),
        ('sC_outer', ctypes.c_int64), ('sC_inner', ctypes.c_int64),
        ('bias', ctypes.c_void_p), ('resid', ctypes.c_void_p), ('ldr', ctypes.c_int64),
        ('aux', ctypes.c_void_p), ('ldaux', ctypes.c_int64),
        ('aux_dtype', ctypes.c_int32), ('epi', ctypes.c_int32), ('accumulate', ctypes.c_int32),
        ('alpha', ctypes.c_float), ('tile', ctypes.c_int32), ('split_k', ctypes.c_int32),
        ('ws', ctypes.c_void_p), ('ws_elems', ctypes.c_int64), ('colsum_part', ctypes.c_void_p),
        ('a_colsum', ctypes.c_void_p), ('a_colsum_acc', ctypes.c_int32),
        ('a_scale', ctypes.c_void_p), ('b_scale', ctypes.c_void_p),
        ('q_out', ctypes.c_void_p), ('q_scale', ctypes.c_void_p), ('ldq', ctypes.c_int64),
        ('a_amax', ctypes.c_void_p),
        ('ln_gamma', ctypes.c_void_p), ('ln_beta', ctypes.c_void_p), ('ln_out', ctypes.c_void_p), ('ln_mean', ctypes.c_void_p),
        ('ln_rstd', ctypes.c_void_p), ('ln_eps', ctypes.c_float),
    ]


_P, _I, _L, _F = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64, ctypes.c_float


class BlockDesc(ctypes.Structure):
    """mirror of mmae_block_desc"""
    _fields_ = ([(n, _I) for n in ('B', 'N', 'D', 'heads', 'Hd', 'act_dtype', 'f32_gemm')] + [('eps', _F)]
                + [(n, _P) for n in ('qkv_w', 'proj_w', 'fc1_w', 'fc2_w',
                                     'n1_w', 'n1_b', 'qkv_b', 'proj_b', 'n2_w', 'n2_b', 'fc1_b', 'fc2_b',
                                     'x0', 'ln1', 'mean1', 'rstd1', 'qkv', 'lse', 'ao', 'x1', 'ln2', 'mean2', 'rstd2', 'hpre', 'hact', 'x2',
                                     'dx', 'dx_act', 'dx0', 'dx0_act',
                                     'd_hpre', 'd_ln2', 'd_ao', 'd_qkv', 'd_ln1', 'dx1', 'dx1_act',
                                     'part_h', 'part1', 'part2',
                                     'g_n1_w', 'g_n1_b', 'g_qkv_w', 'g_qkv_b', 'g_proj_w', 'g_proj_b', 'g_n2_w', 'g_n2_b',
                                     'g_fc1_w', 'g_fc1_b', 'g_fc2_w', 'g_fc2_b', 'g_cs')]
                + [('grad_acc', _I), ('fc2_b_done', _I), ('ws_main', _P), ('ws_main_elems', _L), ('ws_side', _P), ('ws_side_elems', _L)]
                + [(n, _P) for n in ('dp1', 'dp2', 'branch', 'dxs_act')]
                + [('mx_w', _P), ('mx_tmp', _P), ('mx_tmp_bytes', _L)]
                + [('x3_w', _P), ('x3_n', _I), ('x3_tmp', _P), ('x3_tmp_bytes', _L), ('dy_amax', _P)])


class StackDesc(ctypes.Structure):
    """mirror of mmae_stack_desc"""
    _fields_ = ([(n, _I) for n in ('L', 'B', 'N', 'D', 'heads', 'Hd', 'act_dtype', 'f32_gemm')] + [('eps', _F), ('grad_acc', _I)]
                + [(n, _P) for n in ('w', 'p', 'dp', 'x', 'act')] + [('act_bytes', _L)]
                + [(n, _P) for n in ('g', 'd_out', 'dx', 'tmp')] + [('tmp_bytes', _L), ('l_begin', _I), ('l_end', _I)]
                + [('ws_main', _P), ('ws_main_elems', _L), ('ws_side', _P), ('ws_side_elems', _L), ('mx_w', _P)])


class AdapterDesc(ctypes.Structure):
    """mirror of mmae_adapter_desc"""
    _fields_ = ([(n, _I) for n in ('B', 'NC', 'Denc', 'D', 'heads', 'Hd', 'depth', 'T', 'q_task', 'G', 'n_q', 'C', 'nh', 'nw', 'ph', 'pw',
                                   'act_dtype', 'f32_gemm')] + [('eps', _F), ('grad_acc', _I)]
                + [(n, _P) for n in ('task_offsets_host', 'w', 'p', 'mask_token', 'task_emb', 'pos', 'enc', 'enc_act', 'ids_keep',
                                     'ids_restore', 'act')] + [('act_bytes', _L), ('img', _P)]
                + [('d_img', _P), ('d_pat', _P), ('ld_pat', _L), ('g', _P), ('d_enc', _P), ('tmp', _P), ('tmp_bytes', _L)]
                + [('ws_main', _P), ('ws_main_elems', _L), ('ws_side', _P), ('ws_side_elems', _L)]
                + [('x3_w', _P), ('x3_n', _I), ('dy_amax', _P), ('pat', _P)])


class DwProblem(ctypes.Structure):
    """mirror of mmae_dw_problem"""
    _fields_ = [('dy', _P), ('ldy', _L), ('x', _P), ('ldx', _L), ('dw', _P), ('db', _P), ('n_out', _I), ('k_in', _I)]


class DwGroupDesc(ctypes.Structure):
    """mirror of mmae_dw_group_desc"""
    _fields_ = [('n', _I), ('rows', _I), ('ab_dtype', _I), ('accumulate', _I), ('split_k', _I), ('p', DwProblem * 8), ('ws', _P), ('ws_elems', _L), ('unscale', _P)]


class ColsumJob(ctypes.Structure):
    """mirror of mmae_colsum_job"""
    _fields_ = [('src', _P), ('dtype', _I), ('cols', _I), ('rows', _L), ('ld', _L), ('seg_w', _I), ('nseg', _I), ('dst', _P * 8), ('unscale', _P)]


class OptDesc(ctypes.Structure):
    """mirror of mmae_opt_desc"""
    _fields_ = [('p', _P), ('g', _P), ('m', _P), ('v', _P), ('n', _L), ('shadow', _P), ('shadow_dtype', _I),
                ('lr', _F), ('weight_decay', _F), ('beta1', _F), ('beta2', _F), ('eps', _F), ('lrwd_dev', _P),
                ('clip_grad', _F), ('skip_grad', _F), ('grad_prescale', _F), ('loss_dev', _P),
                ('state', _P), ('istate', _P), ('ws', _P), ('found_inf_dev', _P), ('grad_scale_dev', _P)]


class PatchSrc(ctypes.Structure):
    """mirror of mmae_patch_src"""
    _fields_ = [
        ('data', ctypes.c_void_p), ('emb', ctypes.c_void_p),
        ('kind', ctypes.c_int32), ('C', ctypes.c_int32), ('H', ctypes.c_int32), ('W', ctypes.c_int32),
        ('ph', ctypes.c_int32), ('pw', ctypes.c_int32), ('k_off', ctypes.c_int32), ('n_cls', ctypes.c_int32),
    ]


_SCALARS = {
    'int': ctypes.c_int, 'int32_t': ctypes.c_int32, 'int64_t': ctypes.c_int64, 'float': ctypes.c_float,
    'size_t': ctypes.c_size_t,
}
_RET = {'int': ctypes.c_int, 'int64_t': ctypes.c_int64, 'const char*': ctypes.c_char_p}


def _parse_header(path: str) -> Dict[str, Tuple[object, List[object]]]:
    src = open(path).read()
    if not os.environ.get('MMAE_EXPERIMENTS'):           # prototypes that exist in experiment builds of the library only
        src = re.sub(r'#ifdef\s+MMAE_EXPERIMENTS.*?#endif', ' ', src, flags=re.S)
    src = re.sub(r'/\*.*?\*/', ' ', src, flags=re.S)
    src = re.sub(r'//[^\n]*', ' ', src)
    src = re.sub(r'typedef\s+struct\s+\w+\s*\{.*?\}\s*\w+\s*;', ' ', src, flags=re.S)
    protos = {}
    for m in re.finditer(r'(const\s+char\s*\*|int64_t|int)\s+(mmae_\w+)\s*\(([^)]*)\)\s*;', src):
        ret = re.sub(r'\s+', ' ', m.group(1)).replace(' *', '*')
        name, args = m.group(2), m.group(3).strip()
        argtypes = []
        if args and args != 'void':
            for a in args.split(','):
                a = a.strip()
                if '*' in a:
                    argtypes.append(ctypes.c_void_p)
                else:
                    ty = a.split()[-2] if len(a.split()) >= 2 else a
                    argtypes.append(_SCALARS[ty])
        protos[name] = (_RET[ret], argtypes)
    return protos


_PROTOS = _parse_header(HEADER) if os.path.exists(HEADER) else {}
_lib = None


def declared_symbols() -> List[str]:
    return sorted(_PROTOS)


def load() -> ctypes.CDLL:
    """Load libmmae_hip.so (once) and attach prototypes.  Raises if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f'{LIB_PATH} not found: the MI355X kernel library is not built. '
            'Run `python -c "import __graft_entry__ as g; g.build()"` (or `make -C multimae_amd/csrc`). '
            'multimae_amd has no CPU / PyTorch fallback by design.')
    lib = ctypes.CDLL(LIB_PATH)
    for name, (ret, argtypes) in _PROTOS.items():
        fn = getattr(lib, name)          # AttributeError if the .so lacks a declared symbol
        fn.restype = ret
        fn.argtypes = argtypes
    if lib.mmae_abi_version() != 7:
        raise RuntimeError('libmmae_hip.so ABI version mismatch')
    for which, cls in enumerate((GemmDesc, BlockDesc, StackDesc, AdapterDesc, OptDesc, PatchSrc, DwGroupDesc, ColsumJob)):
        if lib.mmae_struct_size(which) != ctypes.sizeof(cls):
            raise RuntimeError(f'{cls.__name__}: ctypes mirror ({ctypes.sizeof(cls)} B) != library struct ({lib.mmae_struct_size(which)} B)')
    if os.environ.get('MMAE_MX_WGRAD') is not None:      # A/B: bf16 (0) or MX-fp8 (1) weight gradients in MX-fp8 mode (ops.mx_wgrad)
        lib.mmae_mx_wgrad(int(os.environ['MMAE_MX_WGRAD'] != '0'))
    if os.environ.get('MMAE_LN_FUSE') is not None:       # A/B: decoder LayerNorms as side outputs of the preceding products (1) or own launches (0)
        lib.mmae_ln_fuse(int(os.environ['MMAE_LN_FUSE'] != '0'))
    if os.environ.get('MMAE_XATTN_FUSE') is not None:    # A/B: the adapters' cross-attention with its two projections inside the attention launch (1) or as three launches (0)
        lib.mmae_xattn_fuse(int(os.environ['MMAE_XATTN_FUSE'] != '0'))
    if os.environ.get('MMAE_GELU_GRAD_AUX') is not None:  # A/B: the MLP pair with the derivative stored by the forward (1) or re-evaluated (0)
        lib.mmae_gelu_grad_aux(int(os.environ['MMAE_GELU_GRAD_AUX'] != '0'))
    _lib = lib
    return lib


class KernelError(RuntimeError):
    pass


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = load().mmae_last_error()
        raise KernelError(f'{what} failed (rc={rc}): {msg.decode() if msg else "?"}')
