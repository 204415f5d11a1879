"""CPU dry-run of the engine's host logic: the C ABI is replaced by a stub that only validates
argument counts/types against include/mmae.h, so the whole Python control flow (buffer shapes,
autograd plumbing, gradient routing) runs here without a GPU.  Numbers are garbage by design."""
import ctypes

import torch

from multimae_amd import _lib, ops


class FakeLib:
    def __getattr__(self, name):
        ret, argtypes = _lib._PROTOS[name]

        def fn(*args):
            assert len(args) == len(argtypes), (name, len(args), len(argtypes))
            for i, (a, t) in enumerate(zip(args, argtypes)):
                if t is ctypes.c_void_p:
                    ok = a is None or isinstance(a, (int, ctypes.c_void_p, ctypes.Array)) or type(a).__name__ == 'CArgObject'
                elif t is ctypes.c_float:
                    ok = isinstance(a, (int, float)) and not isinstance(a, bool)
                else:
                    ok = isinstance(a, int)
                assert ok, (name, i, type(a), t)
            if name == 'mmae_abi_version':
                return 7
            if name == 'mmae_struct_size':
                return ctypes.sizeof((_lib.GemmDesc, _lib.BlockDesc, _lib.StackDesc, _lib.AdapterDesc, _lib.OptDesc, _lib.PatchSrc, _lib.DwGroupDesc, _lib.ColsumJob)[args[0]])
            # slab layouts of the composite entry points: enough room for the views the host code cuts out of them
            if name in ('mmae_stack_act_bytes', 'mmae_stack_out_offset'):
                d = args[0]._obj
                per = d.B * d.N * d.D * 4
                return d.L * per + 256 if name == 'mmae_stack_act_bytes' else args[1] * per
            if name == 'mmae_adapter_act_bytes':
                d = args[0]._obj
                return d.B * d.n_q * d.C * d.ph * d.pw * 4 + 256
            if name in ('mmae_stack_tmp_bytes', 'mmae_adapter_tmp_bytes'):
                return 256
            if name == 'mmae_adapter_pat_offset':
                return 0
            if name == 'mmae_loss_split':
                return 8
            if name == 'mmae_layernorm_bwd_nblk':
                return max(1, min(1024, (args[0] + 3) // 4))
            if name in ('mmae_tokens_assemble_bwd_nblk', 'mmae_decoder_build_bwd_nblk'):
                return min(args[0], 256)
            if name == 'mmae_colsum_ws_elems':
                return 256 * args[1]
            if name == 'mmae_last_error':
                return b'stub'
            return 0
        return fn


def install():
    _lib._lib = FakeLib()
    ops._require_gpu = lambda t, name='tensor': None
    ops._stream = lambda: 0
    ops._device_ok = lambda t: True
    ops._WS_ELEMS[0] = 1024
    from multimae_amd import functions
    functions.ops._require_gpu = ops._require_gpu
