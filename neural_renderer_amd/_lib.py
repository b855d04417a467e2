"""ctypes binding of libnr_hip.so (include/nr_hip.h).  The product path has NO fallback: if the HIP
library is missing or an entry point fails, this raises."""
import ctypes
import os

from . import _build

_c = ctypes
_vp, _i32, _f64, _sz = _c.c_void_p, _c.c_int32, _c.c_double, _c.c_size_t



class Camera(_c.Structure):
    """struct nr_camera (include/nr_hip.h)."""
    _fields_ = [('mode', _i32), ('perspective', _i32), ('target', _c.c_float * 3), ('up', _c.c_float * 3),
                ('width', _c.c_float)]


class Light(_c.Structure):
    """struct nr_light (include/nr_hip.h)."""
    _fields_ = [('intensity_ambient', _c.c_float), ('intensity_directional', _c.c_float),
                ('color_ambient', _c.c_float * 3), ('color_directional', _c.c_float * 3), ('direction', _c.c_float * 3)]


class FaceLight(_c.Structure):
    """struct nr_face_light (include/nr_hip.h): per-face light colours for the _lit entry points."""
    _fields_ = [('light', _vp), ('texture_faces', _i32), ('textures', _vp), ('grad_light', _vp)]


_cam_p, _light_p, _fl_p = _c.POINTER(Camera), _c.POINTER(Light), _c.POINTER(FaceLight)

# name -> (restype, argtypes); mirrors include/nr_hip.h one to one
SIGNATURES = {
    'nr_version': (_c.c_int, []),
    'nr_error_string': (_c.c_char_p, [_c.c_int]),
    'nr_forward_workspace_bytes': (_sz, [_i32, _i32, _i32]),
    'nr_backward_workspace_bytes': (_sz, [_i32, _i32, _i32, _i32, _i32]),
    'nr_forward_face_index_map': (_c.c_int, [_vp] * 6 + [_i32, _i32, _i32, _f64, _f64, _vp, _sz, _vp]),
    'nr_forward_texture_sampling': (_c.c_int, [_vp] * 10 + [_i32, _vp, _i32, _i32, _i32, _i32, _f64, _i32, _vp]),
    'nr_backward_pixel_map': (_c.c_int, [_vp] * 7 + [_i32, _i32, _i32, _f64, _i32, _i32, _i32, _vp, _vp, _sz, _vp]),
    'nr_backward_textures': (_c.c_int, [_vp] * 9 + [_i32, _i32, _i32, _i32, _f64, _i32, _vp]),
    'nr_backward_depth_map': (_c.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _vp]),
    'nr_forward_rasterize': (_c.c_int, [_vp] * 10 + [_i32] * 5 + [_f64] * 3 + [_i32, _vp, _sz, _vp]),
    'nr_vertices_to_faces': (_c.c_int, [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp]),
    'nr_vertices_to_faces_backward': (_c.c_int, [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp]),
    'nr_backward_rasterize': (_c.c_int, [_vp] * 12 + [_i32] * 4 + [_f64, _i32, _vp, _vp, _sz, _vp]),
    'nr_image_epilogue': (_c.c_int, [_vp] * 6 + [_i32] * 3 + [_vp]),
    'nr_image_epilogue_backward': (_c.c_int, [_vp] * 6 + [_i32] * 3 + [_vp]),
    'nr_load_textures': (_c.c_int, [_vp] * 4 + [_i32] * 4 + [_vp]),
    'nr_create_texture_image': (_c.c_int, [_vp] * 3 + [_i32] * 5 + [_vp]),
    'nr_adam_update': (_c.c_int, [_vp] * 4 + [_sz] + [_c.c_float] * 4 + [_vp]),
    'nr_frontend_workspace_bytes': (_sz, [_i32]),
    'nr_frontend_forward': (_c.c_int, [_vp] * 6 + [_i32] * 7 + [_cam_p, _light_p, _vp]),
    'nr_frontend_backward': (_c.c_int, [_vp] * 9 + [_i32] * 7 + [_cam_p, _light_p, _vp, _sz, _vp]),
    'nr_forward_rasterize_lit': (_c.c_int, [_fl_p] + [_vp] * 10 + [_i32] * 5 + [_f64] * 3 + [_i32, _vp, _sz, _vp]),
    'nr_backward_rasterize_lit': (_c.c_int, [_fl_p] + [_vp] * 12 + [_i32] * 4 + [_f64, _i32, _vp, _vp, _sz, _vp]),
    'nr_frontend_forward_light': (_c.c_int, [_vp] * 5 + [_i32] * 6 + [_cam_p, _light_p, _vp]),
    'nr_frontend_backward_light': (_c.c_int, [_vp] * 7 + [_i32] * 6 + [_cam_p, _light_p, _vp, _sz, _vp]),
}

NR_VERSION = 600  # include/nr_hip.h; load() refuses a library of another version (a stale build)
NR_FLAG_FIX_TEXTURE_BATCH_Z = 1
NR_FLAG_EXACT_GRADIENT = 2
NR_FLAG_K6_GLOBAL = 4
NR_FLAG_K6_SCAN = 8
NR_FLAG_ZBUF_EPOCH = 16  # + epoch number << 8 (include/nr_hip.h)
NR_FLAG_SPARSE_WEIGHT_MAP = 32
NR_FLAG_SERIAL_BACKWARD = 64
NR_FLAG_K6_LEGACY = 128
NR_FLAG_K6_PX = 65536  # (accepted and ignored since 0.6.0; bits 8..15 of a forward's flags carry the epoch number)
NR_E_INDEX = -6
NR_CAMERA_LOOK_AT = 1
NR_CAMERA_LOOK = 2

_lib = None


class NRError(RuntimeError):
    pass


def load():
    """Load (never build implicitly at import on a GPU box: the .so ships in-tree; build() exists for that)."""
    global _lib
    if _lib is not None:
        return _lib
    path = os.environ.get('NR_HIP_LIB') or _build.LIB_PATH  # NR_HIP_LIB: a variant build (development timing runs)
    if not os.path.exists(path):
        raise NRError('%s not found: build it with `python -m neural_renderer_amd._build` '
                      '(or __graft_entry__.build()); there is no CPU/eager fallback.' % path)
    lib = _c.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the library does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    if lib.nr_version() != NR_VERSION and not os.environ.get('NR_HIP_LIB'):  # (a development variant: the caller's business)
        raise NRError('%s is version %d, this package binds version %d: rebuild it (python -m neural_renderer_amd._build)'
                      % (path, lib.nr_version(), NR_VERSION))
    _lib = lib
    return lib


# the measurement build's two extra exports (include/nr_hip_profile.h)
PROFILE_SIGNATURES = {
    'nr_profile_band_kernel': (_c.c_int, [_i32]),
    'nr_profile_band_kernel_ms': (_c.c_float, []),
    'nr_profile_band_kernel_which': (_c.c_int, []),
    'nr_profile_k6_choice': (_c.c_int, [_i32, _i32, _i32, _i32, _i32, _c.c_double, _i32]),
}
_profile_lib = None


def load_profile():
    """libnr_hip_prof.so -- the product's sources with the band-kernel timing hook compiled in -- for measurements only
    (bench.py's roofline, the hook's test).  The product path never loads it."""
    global _profile_lib
    if _profile_lib is None:
        path = _build.PROFILE_LIB_PATH
        if not os.path.exists(path):
            raise NRError('%s not found: build it with neural_renderer_amd._build.build_profile()' % path)
        lib = _c.CDLL(path)
        for name, (res, args) in list(SIGNATURES.items()) + list(PROFILE_SIGNATURES.items()):
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        if lib.nr_version() != NR_VERSION:
            raise NRError('%s is version %d, this package binds version %d: rebuild it' % (path, lib.nr_version(), NR_VERSION))
        _profile_lib = lib
    return _profile_lib


def check(code, what):
    if code != 0:
        msg = load().nr_error_string(code)
        text = '%s failed (%d): %s' % (what, code, msg.decode() if msg else '?')
        if code == NR_E_INDEX:
            raise IndexError(text)
        raise NRError(text)


def ptr(t):
    """data_ptr of a tensor or None."""
    return None if t is None else t.data_ptr()
