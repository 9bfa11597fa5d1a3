"""ctypes binding of libkimg.so (the C ABI declared in include/kimg.h).

The HIP extension is mandatory: there is no CPU fallback.  ``lib()`` raises
:class:`KimgLibraryError` if the shared library has not been built or cannot
be loaded.  PyTorch-ROCm is imported first so that libkimg.so resolves the HIP
runtime and rocFFT already loaded by torch (same SONAMEs) -- device pointers
and streams created by torch are then valid inside the library.
"""
import ctypes
import threading
import os
from ctypes import c_int, c_int64, c_float, c_double, c_size_t, c_uint32, c_void_p, c_char_p

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, 'libkimg.so')

P = c_void_p
I = c_int
L = c_int64
F = c_float

VERSION = 4     # KIMG_VERSION of include/kimg.h

PROTOTYPES = {
    'kimg_version': (c_int, []),
    'kimg_error_string': (c_char_p, [I]),
    'kimg_kernel_table': (c_int, [P, P, I, I, I, I, c_double, c_double, c_double, P]),
    'kimg_grid_workspace_bytes': (c_size_t, [L, I, I, I, I]),
    'kimg_grid_binned_workspace_bytes': (c_size_t, [L, I, I, I, I]),
    'kimg_grid_jumps': (c_int, [P, L, I, P, P]),
    'kimg_grid': (c_int, [P, L, L, I, I, P, L, L, P, P, P, L, P, I, I, I, P, c_size_t, I, I, P]),
    'kimg_degrid': (c_int, [P, L, L, I, I, P, P, P, P, L, P, I, I, I, P, c_size_t, I, I, P]),
    'kimg_degrid_workspace_bytes': (c_size_t, [I, I, I, I]),
    'kimg_degrid_binned_workspace_bytes': (c_size_t, [L, I, I, I, I]),
    'kimg_predict': (c_int, [P, P, P, P, P, P, L, I, I, I, F, F, F, P]),
    'kimg_grid_weights': (c_int, [P, L, L, I, I, I, P, P, L, P]),
    'kimg_mean_weight': (c_int, [P, P, L, I, I, P]),
    'kimg_density_weights': (c_int, [P, P, L, L, I, I, I, F, F, P]),
    'kimg_density_weights_robust': (c_int, [P, P, L, L, I, I, I, P, c_double, F, P]),
    'kimg_fill': (c_int, [P, L, F, P]),
    'kimg_preprocess_convert': (c_int, [I, I, L, P, P, P, P, P, P, P, F, I, I, I, F, P, P, P, P]),
    'kimg_preprocess_workspace_bytes': (ctypes.c_size_t, [L, I]),
    'kimg_preprocess_compress': (c_int, [I, L, I, P, P, P, P, P, P, P, P, L, P, ctypes.c_size_t, P]),
    'kimg_real_to_complex': (c_int, [P, P, L, P]),
    'kimg_store_reorder_workspace_bytes': (c_size_t, [L]),
    'kimg_store_reorder': (c_int, [I, L, I, I, I, I, P, P, P, P, P, P, P, P, P, P, c_size_t, P]),
    'kimg_grid_to_layer': (c_int, [P, I, P, L, I, P]),
    'kimg_grid_to_half_layer': (c_int, [P, I, P, L, I, P]),
    'kimg_real_layer_to_image': (c_int, [P, L, P, L, I, P, F, F, P]),
    'kimg_image_to_real_layer': (c_int, [P, L, P, L, I, P, F, F, P]),
    'kimg_half_layer_to_grid': (c_int, [P, L, I, P, I, P]),
    'kimg_set_window_cus': (c_int, [I]),
    'kimg_get_window_cus': (c_int, []),
    'kimg_grid_image_real_supported': (c_int, [I, I]),
    'kimg_grid_image_real_workspace_bytes': (c_size_t, [I, I]),
    'kimg_grid_to_image_real': (c_int, [P, L, I, P, L, I, P, F, F, I, P, c_size_t, P]),
    'kimg_image_to_grid_real': (c_int, [P, L, I, P, L, I, P, F, F, P, c_size_t, P]),
    'kimg_convolve_beam': (c_int, [P, L, I, F, F, F, F, P, c_size_t, P]),
    'kimg_grid_image_w_workspace_bytes': (c_size_t, [I, I]),
    'kimg_grid_to_image_w': (c_int, [P, L, I, P, L, I, P, F, F, F, I, P, c_size_t, P]),
    'kimg_image_to_grid_w': (c_int, [P, L, I, P, L, I, P, F, F, F, P, c_size_t, P]),
    'kimg_layer_to_grid': (c_int, [P, L, I, P, I, P]),
    'kimg_layer_to_image': (c_int, [P, L, P, I, P, F, F, F, P]),
    'kimg_image_to_layer': (c_int, [P, P, L, I, P, F, F, F, P]),
    'kimg_fft_plan_create': (c_int, [ctypes.POINTER(c_void_p), I, I]),
    'kimg_fft_exec': (c_int, [P, P, I, P]),
    'kimg_fft_plan_destroy': (c_int, [P]),
    'kimg_rfft_plan_create': (c_int, [ctypes.POINTER(c_void_p), I, I]),
    'kimg_rfft_exec': (c_int, [P, P, P, I, P]),
    'kimg_rfft_plan_destroy': (c_int, [P]),
    'kimg_fourier_beam': (c_int, [P, L, I, I, F, F, F, F, P]),
    'kimg_image_peak': (c_int, [P, L, L, P, L, I, I, I, F, P, P]),
    'kimg_image_nansum': (c_int, [P, L, L, I, I, I, P, P]),
    'kimg_scale': (c_int, [P, L, L, I, I, I, ctypes.POINTER(c_float), P]),
    'kimg_pixel_reciprocal': (c_int, [P, L, L, I, I, I, I, I, P, P]),
    'kimg_scale_device': (c_int, [P, L, L, I, I, I, P, P]),
    'kimg_add_image': (c_int, [P, L, L, P, L, L, I, I, I, P]),
    'kimg_apply_primary_beam': (c_int, [P, L, L, P, L, I, I, I, F, F, P]),
    'kimg_psf_patch': (c_int, [P, L, L, I, I, I, I, I, I, I, F, P, P]),
    'kimg_abs_histogram': (c_int, [P, L, L, I, I, I, I, I, c_uint32, P, P]),
    'kimg_abs_count_le': (c_int, [P, L, L, I, I, I, I, F, P, P]),
    'kimg_update_tiles': (c_int, [P, L, L, I, I, I, I, I, P, P, I, I, I, I, I, I, P]),
    'kimg_find_peak': (c_int, [P, L, L, I, P, P, I, I, P, P, P, P]),
    'kimg_subtract_psf': (c_int, [P, P, L, L, I, I, I, P, L, L, I, I, I, I, P, I, I, F, P]),
    'kimg_noise_est_scratch_bytes': (c_size_t, []),
    'kimg_preload': (c_int, []),
    'kimg_noise_est': (c_int, [P, L, L, I, I, I, I, F, P, P, P]),
    'kimg_clean_state_bytes': (c_size_t, [I, I, I]),
    'kimg_clean_cycles': (c_int, [P, P, L, L, I, I, I, P, L, L, I, I, I, I, I, I, F, F,
                                  P, P, I, I, I, I, P, P, P]),
    'kimg_clean_major_cycles': (c_int, [P, P, L, L, I, I, I, P, L, L, I, I, I, I, I, I, F, c_double, c_double,
                                        P, P, I, I, I, I, P, P, P, P, P]),
    'kimg_clean_cycles_batch': (c_int, [P, I, L, L, I, I, I, L, L, I, I, I, I, F, I, I, P]),
}


class CleanChannel(ctypes.Structure):
    """``kimg_clean_channel`` of include/kimg.h (one channel of kimg_clean_cycles_batch)."""
    _fields_ = [('dirty', c_void_p), ('model', c_void_p), ('psf', c_void_p),
                ('tile_max', c_void_p), ('tile_pos', c_void_p), ('state', c_void_p),
                ('log', c_void_p), ('patch_width', ctypes.c_int32),
                ('patch_height', ctypes.c_int32), ('threshold', c_float),
                ('max_cycles', ctypes.c_int32)]


CLEAN_BATCH_MAX = 8     # KIMG_CLEAN_BATCH_MAX


class KimgLibraryError(RuntimeError):
    pass


class KimgError(RuntimeError):
    def __init__(self, code, what):
        super().__init__('{} failed: {} ({})'.format(what, error_string(code), code))
        self.code = code


_lib = None


_load_lock = threading.Lock()


def lib():
    """Load libkimg.so once; raise KimgLibraryError if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    with _load_lock:
        return _load()


def preload():
    """``kimg_preload`` (include/kimg.h): every code object of the library loaded on the current
    device now, in this thread -- before several threads make their first launches at once."""
    check(lib().kimg_preload(), 'kimg_preload')


def _load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise KimgLibraryError(
            'libkimg.so has not been built (run `python -m katsdpimager_amd.build`); '
            'katsdpimager_amd has no CPU fallback')
    try:
        import torch  # noqa: F401  (loads torch's libamdhip64 / librocfft first)
    except ImportError:
        pass
    try:
        handle = ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_GLOBAL)
    except OSError as e:
        raise KimgLibraryError('cannot load {}: {}'.format(LIB_PATH, e)) from e
    for name, (restype, argtypes) in PROTOTYPES.items():
        try:
            fn = getattr(handle, name)
        except AttributeError as e:
            raise KimgLibraryError('libkimg.so lacks symbol ' + name) from e
        fn.restype = restype
        fn.argtypes = argtypes
    if handle.kimg_version() != VERSION:
        raise KimgLibraryError('libkimg.so version mismatch')
    _lib = handle
    return _lib


def error_string(code):
    return lib().kimg_error_string(code).decode()


def check(code, what):
    if code != 0:
        raise KimgError(code, what)
