"""ctypes binding of libhyperreel_hip.so (include/hyperreel_hip.h).

There is deliberately no fallback: if the library is missing or a call fails this module
raises.  The product never routes through PyTorch ops or the CPU oracle.
"""
import ctypes as C
import os

from .plan import hr_camera, hr_config, hr_fields

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, '_build', 'libhyperreel_hip.so')

ABI_VERSION = 26



HR_OPT_FRAME_KERNEL, HR_OPT_SAMPLE_WAVES, HR_OPT_FRAME_KERNEL_ACTIVE, HR_OPT_MLP_PRECISION_ACTIVE, HR_OPT_MLP_OVERFLOW, HR_OPT_MLP_CALIBRATED = 0, 1, 2, 3, 4, 5
HR_OPT_TRAIN_DETERMINISTIC = 6
HR_OPT_MLP_F8_SATURATED = 7
HR_OPT_MLP_VERIFIED, HR_OPT_REDO_COUNT, HR_OPT_REDO_OVERFLOW, HR_OPT_WIDE_COUNT, HR_OPT_CHUNK_RAYS = 8, 9, 10, 11, 12
HR_E_RANGE = -5


class hr_train_tensors(C.Structure):
    """Device pointers of the trainable tensors (or of their gradients), reference layouts (include/hyperreel_hip.h)."""
    _fields_ = [('density_a', C.c_void_p * 3), ('density_b', C.c_void_p * 3), ('app_a', C.c_void_p * 3), ('app_b', C.c_void_p * 3),
                ('basis', C.c_void_p), ('color_table', C.c_void_p)]


class hr_verify_info(C.Structure):
    """What the verified fast path rests on for one model (include/hyperreel_hip.h)."""
    _fields_ = [('verified', C.c_int32), ('fallback', C.c_int32), ('band', C.c_float), ('band_q', C.c_float), ('band_off', C.c_float),
                ('band_floor', C.c_float), ('max_d_zc', C.c_float), ('max_d_dist_n', C.c_float), ('max_d_geo_n', C.c_float), ('max_d_off', C.c_float),
                ('max_d_dist', C.c_float), ('max_d_head', C.c_float), ('listed_frac', C.c_float), ('max_d_rgb', C.c_float),
                ('n_rays', C.c_int64), ('n_rays_used', C.c_int64), ('n_samples', C.c_int64), ('n_flipped', C.c_int64), ('n_shaky', C.c_int64)]


# every symbol include/hyperreel_hip.h declares: (name, restype, argtypes)
SYMBOLS = [
    ('hr_abi_version', C.c_int, []),
    ('hr_sizeof_config', C.c_int, []),
    ('hr_last_error', C.c_char_p, []),
    ('hr_model_create', C.c_int, [C.POINTER(hr_config), C.POINTER(C.c_void_p)]),
    ('hr_model_create_cascade', C.c_int, [C.POINTER(hr_config), C.POINTER(hr_config), C.POINTER(C.c_void_p)]),
    ('hr_model_upload', C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, C.c_size_t]),
    ('hr_model_finalize', C.c_int, [C.c_void_p]),
    ('hr_model_calibrate', C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.POINTER(C.c_float), C.c_void_p]),
    ('hr_model_update_config', C.c_int, [C.c_void_p, C.POINTER(hr_config), C.c_void_p]),
    ('hr_model_reserve', C.c_int, [C.c_void_p, C.c_int64]),
    ('hr_model_set_option', C.c_int, [C.c_void_p, C.c_int32, C.c_int32]),
    ('hr_model_get_option', C.c_int, [C.c_void_p, C.c_int32, C.POINTER(C.c_int32)]),
    ('hr_model_verify_info', C.c_int, [C.c_void_p, C.POINTER(hr_verify_info)]),
    ('hr_model_set_occupancy', C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_float), C.c_void_p]),
    ('hr_render', C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    ('hr_render_frame', C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_float, C.c_void_p, C.c_void_p]),
    ('hr_render_fields', C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.POINTER(hr_fields), C.c_void_p]),
    ('hr_allgather_tiles', C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    ('hr_shard_range', C.c_int, [C.c_int64, C.c_int32, C.c_int32, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    ('hr_generate_rays', C.c_int, [C.POINTER(hr_camera), C.c_int32, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p]),
    ('hr_upsample_plane', C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]),
    ('hr_train_features', C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    ('hr_train_rows_forward', C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    ('hr_train_rows_backward', C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]),
    ('hr_train_forward', C.c_int, [C.c_void_p, C.POINTER(hr_train_tensors), C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p,
                                   C.c_void_p]),
    ('hr_train_forward_fields', C.c_int, [C.c_void_p, C.POINTER(hr_train_tensors), C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p,
                                          C.POINTER(hr_fields), C.c_void_p]),
    ('hr_train_backward', C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p,
                                    C.POINTER(hr_train_tensors), C.c_void_p]),
    ('hr_dense_alpha', C.c_int, [C.c_void_p, C.POINTER(C.c_int32), C.c_float, C.c_int32, C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_float),
                                 C.c_void_p, C.c_void_p]),
    ('hr_pack_display', C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    ('hr_plane_reg_forward', C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    ('hr_plane_reg_backward', C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    ('hr_adam_step', C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]),
    ('hr_mlp_train_forward', C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_void_p, C.c_int64, C.POINTER(C.c_void_p),
                                      C.POINTER(C.c_int64), C.POINTER(C.c_int32), C.c_void_p, C.c_void_p]),
    ('hr_linear_workspace', C.c_size_t, [C.c_int64, C.c_int32, C.c_int32]),
    ('hr_linear_forward', C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_float, C.c_void_p,
                                    C.c_int64, C.c_void_p]),
    ('hr_linear_backward', C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int64, C.c_int32,
                                     C.c_int32, C.c_float, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    ('hr_stage_mlp', C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    ('hr_stage_samples', C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    ('hr_debug_trace_mlp', C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    ('hr_model_device_bytes', C.c_int64, [C.c_void_p]),
    ('hr_model_destroy', None, [C.c_void_p]),
]

_lib = None


class HipLibraryError(RuntimeError):
    pass


def load():
    """Loads (once) and returns the ctypes handle.  Raises HipLibraryError when the
    in-tree library has not been built (`python -m hyperreel_amd.build`)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HipLibraryError(
            f'{LIB_PATH} is missing: build it with `python -m hyperreel_amd.build` '
            '(hipcc --offload-arch=gfx950).  hyperreel_amd has no CPU or PyTorch fallback.')
    try:
        lib = C.CDLL(LIB_PATH)
    except OSError as e:
        raise HipLibraryError(f'cannot load {LIB_PATH}: {e}') from e
    for name, res, args in SYMBOLS:
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise HipLibraryError(f'{LIB_PATH} does not export {name}') from e
        fn.restype = res
        fn.argtypes = args
    v = lib.hr_abi_version()
    if v != ABI_VERSION:
        raise HipLibraryError(f'ABI version mismatch: library {v}, binding {ABI_VERSION}')
    if lib.hr_sizeof_config() != C.sizeof(hr_config):
        raise HipLibraryError('hr_config layout differs between the library and plan.py')
    _lib = lib
    return lib


class HipRangeError(RuntimeError):
    """HR_E_RANGE: a forced fp16 MLP arithmetic does not fit the model's activation range."""


def check(rc, what):
    if rc != 0:
        msg = load().hr_last_error()
        cls = HipRangeError if rc == HR_E_RANGE else RuntimeError
        raise cls(f'{what} failed (code {rc}): {msg.decode() if msg else "?"}')
