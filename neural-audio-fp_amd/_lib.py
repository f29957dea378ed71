"""ctypes binding of libnafp.so (include/nafp.h).

There is no CPU fallback: if the shared library is missing or a call fails, this
module raises.  `import torch` happens first on purpose: the HIP runtime torch
ships (soname libamdhip64.so.7) is then the one libnafp binds to, so device
pointers and streams are shared between torch and the library.
"""
import ctypes
import os

import torch  # noqa: F401  (must be loaded before libnafp; see module docstring)

_HERE = os.path.dirname(os.path.abspath(__file__))
# NAFP_LIB: load another build of the same ABI (kernel ablation experiments, tools/ablate.sh)
LIB_PATH = os.environ.get('NAFP_LIB') or os.path.join(_HERE, 'libnafp.so')

c_i64 = ctypes.c_int64
c_int = ctypes.c_int
c_float = ctypes.c_float
c_void_p = ctypes.c_void_p

# name -> (restype, argtypes); mirrors include/nafp.h one to one.
PROTOTYPES = {
    'nafp_abi_version': (c_int, []),
    'nafp_status_string': (ctypes.c_char_p, [c_int]),
    'nafp_last_hip_error': (c_int, []),
    'nafp_crc32c_host': (ctypes.c_uint32, [c_void_p, c_i64, ctypes.c_uint32]),
    'nafp_mel_filterbank_host': (c_int, [c_int, c_int, c_int, c_float, c_float, c_void_p]),
    'nafp_melspec_create': (c_int, [ctypes.POINTER(c_void_p), c_int, c_int, c_int, c_int, c_int, c_float, c_float]),
    'nafp_melspec_destroy': (c_int, [c_void_p]),
    'nafp_melspec_n_frames': (c_int, [c_void_p]),
    'nafp_melspec_n_mels': (c_int, [c_void_p]),
    'nafp_melspec_forward_f32': (c_int, [c_void_p, c_void_p, c_i64, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    'nafp_melspec_forward_i16': (c_int, [c_void_p, c_void_p, c_i64, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    'nafp_melspec_finish': (c_int, [c_void_p, c_void_p, c_void_p, c_i64, c_int, c_int, c_void_p]),
    'nafp_melspec_forward_windows_i16': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_i64, c_int, c_int, c_void_p,
                                                 c_void_p, c_void_p]),
    'nafp_encoder_create': (c_int, [ctypes.POINTER(c_void_p), c_int, c_int, c_int]),
    'nafp_encoder_create_ex': (c_int, [ctypes.POINTER(c_void_p), c_int, c_int, c_int, c_int]),
    'nafp_encoder_norm': (c_int, [c_void_p]),
    'nafp_encoder_n_trainable': (c_int, [c_void_p]),
    'nafp_encoder_destroy': (c_int, [c_void_p]),
    'nafp_encoder_n_tensors': (c_int, [c_void_p]),
    'nafp_encoder_tensor_numel': (c_i64, [c_void_p, c_int]),
    'nafp_encoder_tensor_shape': (c_int, [c_void_p, c_int, ctypes.POINTER(c_i64)]),
    'nafp_encoder_flat_dim': (c_i64, [c_void_p]),
    'nafp_encoder_set_weights': (c_int, [c_void_p, ctypes.POINTER(c_void_p), c_void_p]),
    'nafp_encoder_workspace_bytes': (c_i64, [c_void_p, c_i64]),
    'nafp_encoder_forward': (c_int, [c_void_p, c_void_p, c_i64, c_void_p, c_i64, c_void_p, c_void_p, c_int, c_void_p]),
    'nafp_encoder_forward_raw': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_i64, c_void_p, c_i64, c_void_p, c_void_p,
                                         c_int, c_void_p]),
    'nafp_encoder_profile_enable': (c_int, [c_void_p, c_int]),
    'nafp_encoder_profile_count': (c_int, [c_void_p]),
    'nafp_encoder_profile_coarse': (c_int, [c_void_p, c_int]),
    'nafp_encoder_profile_read': (c_int, [c_void_p, c_int, c_void_p]),
    'nafp_conv_timeline': (c_int, [c_void_p, c_i64, c_int, c_int, c_int]),
    'nafp_conv_timeline_grid': (c_int, [c_void_p]),
    'nafp_encoder_set_option': (c_int, [c_void_p, c_int, c_int]),
    'nafp_encoder_train_workspace_bytes': (c_i64, [c_void_p, c_i64]),
    'nafp_encoder_forward_train': (c_int, [c_void_p, c_void_p, c_i64, c_void_p, c_i64, c_void_p, c_int, c_void_p]),
    'nafp_encoder_backward': (c_int, [c_void_p, c_void_p, c_void_p, c_i64, c_void_p, c_i64,
                                      ctypes.POINTER(c_void_p), c_int, c_void_p]),
    'nafp_encoder_grad_group_range': (c_int, [c_void_p, c_int, ctypes.POINTER(c_int), ctypes.POINTER(c_int)]),
    'nafp_encoder_grad_group_wait': (c_int, [c_void_p, c_int, c_void_p]),
    'nafp_encoder_div_enc': (c_int, [c_void_p, c_void_p, c_i64, c_void_p, c_int, c_void_p]),
    'nafp_ntxent_workspace_bytes': (c_i64, [c_i64, c_i64]),
    'nafp_ntxent_forward': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_i64, c_i64, c_i64, c_int,
                                    c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_i64, c_void_p]),
    'nafp_specaug_apply': (c_int, [c_void_p, c_i64, c_int, c_int, c_void_p, c_int, c_void_p, c_float, c_void_p]),
    'nafp_specaug_apply_fill_dev': (c_int, [c_void_p, c_i64, c_int, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p]),
    'nafp_specaug_mean_workspace_bytes': (c_i64, []),
    'nafp_specaug_mean': (c_int, [c_void_p, c_i64, c_void_p, c_void_p, c_i64, c_void_p]),
    'nafp_specaug_apply_ex': (c_int, [c_void_p, c_i64, c_int, c_int, c_void_p, c_int, c_i64, c_void_p, c_void_p, c_i64, c_void_p, c_void_p]),
    'nafp_specaug_range': (c_int, [c_void_p, c_i64, c_void_p, c_void_p, c_i64, c_void_p]),
    'nafp_l2_normalize_rows': (c_int, [c_void_p, c_i64, c_int, c_void_p, c_void_p]),
    'nafp_pack_embedding_grads': (c_int, [c_void_p, c_void_p, c_void_p, c_float, c_i64, c_i64, c_int, c_void_p, c_void_p]),
    'nafp_cosine_decay_lr_host': (c_float, [c_float, c_i64, c_i64, c_float]),
    'nafp_cosine_decay_restarts_lr_host': (c_float, [c_float, c_i64, c_i64, c_float, c_float, c_float]),
    'nafp_adam_step': (c_int, [c_void_p, c_int, c_float, c_float, c_float, c_float, c_i64, c_void_p]),
    'nafp_lamb_workspace_bytes': (c_i64, [c_void_p, c_int]),
    'nafp_triplet_workspace_bytes': (c_i64, [c_i64, c_i64]),
    'nafp_triplet_forward': (c_int, [c_void_p, c_void_p, c_i64, c_i64, c_int, c_int, c_float, c_void_p, c_void_p, c_void_p,
                                     c_void_p, c_void_p, c_i64, c_void_p]),
    'nafp_augment_rows': (c_int, [c_void_p, c_void_p, c_i64, c_int, c_void_p, c_void_p]),
    'nafp_search_index_aux_floats': (c_i64, [c_i64]),
    'nafp_search_index_prepare': (c_int, [c_void_p, c_i64, c_int, c_void_p, c_void_p]),
    'nafp_search_workspace_bytes': (c_i64, [c_i64, c_i64, c_int]),
    'nafp_search_topk_l2': (c_int, [c_void_p, c_i64, c_void_p, c_void_p, c_i64, c_int, c_int, c_void_p, c_void_p,
                                    c_void_p, c_i64, c_void_p]),
    'nafp_search_seq_scores': (c_int, [c_void_p, c_void_p, c_i64, c_int, c_void_p, c_void_p, c_i64, c_void_p, c_int,
                                       c_void_p, c_void_p]),
    'nafp_minisearch_scores': (c_int, [c_void_p, c_void_p, c_i64, c_i64, c_int, c_int, c_void_p, c_void_p]),
    'nafp_minisearch_ranks': (c_int, [c_void_p, c_i64, c_i64, c_int, c_int, c_int, c_void_p, c_void_p]),
    'nafp_lamb_step': (c_int, [c_void_p, c_int, c_float, c_float, c_float, c_float, c_float, c_i64,
                               c_void_p, c_i64, c_void_p]),
}


class OptTensor(ctypes.Structure):
    """nafp_opt_tensor (include/nafp.h)."""
    _fields_ = [('param', c_void_p), ('grad', c_void_p), ('m', c_void_p), ('v', c_void_p),
                ('numel', c_i64), ('var_len', c_i64)]


class Rect(ctypes.Structure):
    """nafp_rect (include/nafp.h): inclusive hole bounds."""
    _fields_ = [('f0', c_int), ('f1', c_int), ('t0', c_int), ('t1', c_int)]


# nafp_aug_row (include/nafp.h) as a numpy structured dtype: 64 bytes per row
AUG_ROW_DTYPE = [('ev_off', '<i8'), ('nz_off', '<i8'), ('nz2_off', '<i8'), ('ir_off', '<i8'), ('ev_valid', '<i4'),
                 ('nz_valid', '<i4'), ('nz2_valid', '<i4'), ('ir_len', '<i4'), ('snr_db', '<f4'), ('amp', '<f4'),
                 ('mix', '<i4'), ('reserved', '<i4')]


class NafpError(RuntimeError):
    pass


_lib = None


def load():
    """Load libnafp.so (once).  Raises NafpError if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NafpError(
            f'{LIB_PATH} is missing: the HIP extension is not built. '
            'Run `python neural-audio-fp_amd/build.py` (or __graft_entry__.build()). '
            'There is no CPU fallback for this path.')
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)       # AttributeError if a declared symbol is not exported
        fn.restype = res
        fn.argtypes = args
    if lib.nafp_abi_version() != 1:
        raise NafpError('libnafp ABI version mismatch')
    _lib = lib
    return lib


def check(status, what=''):
    if status != 0:
        lib = load()
        msg = lib.nafp_status_string(status).decode()
        extra = ''
        if status == 3:
            extra = f' (hipError_t={lib.nafp_last_hip_error()})'
        raise NafpError(f'libnafp {what}: {msg}{extra}')


def ptr(t):
    """Device pointer of a torch tensor (or None)."""
    return None if t is None else c_void_p(t.data_ptr())


def current_stream():
    return c_void_p(torch.cuda.current_stream().cuda_stream)


def require_cuda(t, name='tensor'):
    if not t.is_cuda:
        raise NafpError(f'{name} must live on the GPU (got {t.device}); this path has no CPU fallback')
    return t
