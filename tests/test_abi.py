"""CPU checks of the C-ABI library: it builds for gfx950, loads, and exports every symbol that
include/nr_hip.h declares with the signature table the Python binding uses.  No compute calls."""
import ctypes
import os
import re

import pytest

from neural_renderer_amd import _build, _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, 'include', 'nr_hip.h')


def declared_functions():
    src = open(HEADER).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(nr_[a-z_0-9]+)\s*\(', src)))


@pytest.fixture(scope='module')
def lib_path():
    _build.build_profile()  # (the measurement build, with the same staleness check: tests below read the kernel choice through it)
    return _build.build()


def declared_prototypes():
    """name -> list of C parameter types as include/nr_hip.h declares them ('ptr', 'i32', 'f64', 'f32', 'size')."""
    src = open(HEADER).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    out = {}
    for name, params in re.findall(r'\b(nr_[a-z_0-9]+)\s*\(([^)]*)\)\s*;', src):
        kinds = []
        for prm in [x.strip() for x in params.split(',')]:
            if prm in ('void', ''):
                continue
            if '*' in prm:
                kinds.append('ptr')
            elif prm.startswith('int32_t') or prm.startswith('int '):
                kinds.append('i32')
            elif prm.startswith('double'):
                kinds.append('f64')
            elif prm.startswith('float'):
                kinds.append('f32')
            elif prm.startswith('size_t'):
                kinds.append('size')
            else:
                raise AssertionError('unparsed parameter %r of %s' % (prm, name))
        out[name] = kinds
    return out


def test_header_and_binding_agree():
    names = declared_functions()
    assert 'nr_forward_face_index_map' in names and 'nr_backward_pixel_map' in names
    assert sorted(_lib.SIGNATURES) == names
    # every parameter of every prototype, position by position, against the ctypes table the product binds with
    kind_of = {ctypes.c_void_p: 'ptr', ctypes.c_int32: 'i32', ctypes.c_int: 'i32', ctypes.c_double: 'f64',
               ctypes.c_float: 'f32', ctypes.c_size_t: 'size', ctypes.c_char_p: 'ptr'}
    protos = declared_prototypes()
    assert sorted(protos) == names
    for name, (_, args) in _lib.SIGNATURES.items():
        got = [kind_of.get(a, 'ptr') for a in args]  # POINTER(struct) types count as pointers
        assert got == protos[name], (name, got, protos[name])


def test_library_exports_every_declared_symbol(lib_path):
    lib = ctypes.CDLL(lib_path)
    for name in declared_functions():
        assert hasattr(lib, name), name


def test_host_only_entry_points(lib_path):
    lib = _lib.load()
    assert lib.nr_version() == 600 == _lib.NR_VERSION
    import neural_renderer_amd
    v = _lib.NR_VERSION
    assert neural_renderer_amd.__version__ == '%d.%d.%d' % (v // 1000, v // 100 % 10, v % 100)
    assert lib.nr_error_string(0) == b'success'
    assert b'workspace' in lib.nr_error_string(-3)
    assert lib.nr_forward_workspace_bytes(64, 4928, 256) >= 64 * 256 * 256 * 8 + 64 * 4928 * 4
    assert lib.nr_forward_workspace_bytes(0, 1, 1) == 0
    assert lib.nr_forward_workspace_bytes(1, 1, 20000) == 0
    # the measurement hook is not part of the product library (include/nr_hip_profile.h: libnr_hip_prof.so only)
    assert not any(hasattr(lib, n) for n in ('nr_profile_band_kernel', 'nr_profile_band_kernel_ms', 'nr_profile_band_kernel_which',
                                             'nr_profile_k6_choice'))


def test_k6_band_kernel_choice_table(lib_path):
    """Which band kernel a K6 call takes (k6_row_band, read through the measurement build: host logic, no device).  Both arithmetic
    modes have one band kernel since 0.6.0 -- k_bpm_row, ahead of k_bpm_fast on every shape of profiles/r06_k6_kernels.md --, so the
    call's batch size must not enter: a batch and its shards take the same kernel (tests/test_full_size_gpu.py checks the results)."""
    choice = _lib.load_profile().nr_profile_k6_choice
    ROW, FAST, T = 1, 0, 4928  # (teapot with fill_back)
    table = [
        # B, F, S, rgb, alpha, eps, flags -> kernel
        ((64, T, 256, 1, 1, 1e-3, 0), ROW), ((64, T, 256, 0, 1, 1e-3, 0), ROW), ((64, T, 256, 1, 0, 1e-3, 0), ROW),
        ((128, T, 256, 1, 1, 1e-3, 0), ROW), ((32, T, 256, 1, 1, 1e-3, 0), ROW), ((16, T, 256, 1, 0, 1e-3, 0), ROW),
        ((8, T, 256, 0, 1, 1e-3, 0), ROW), ((1, T, 256, 1, 1, 1e-3, 0), ROW),           # shards: the same kernel as the batch
        ((64, T, 512, 1, 1, 1e-3, 0), ROW), ((64, T, 384, 1, 1, 1e-3, 0), ROW), ((64, T, 640, 1, 0, 1e-3, 0), ROW),
        ((64, T, 1024, 1, 0, 1e-3, 0), ROW), ((4, T, 1024, 1, 1, 1e-3, 0), ROW), ((3, 40, 33, 1, 1, 1e-3, 0), ROW),
        ((64, 10240, 256, 1, 0, 1e-3, 0), ROW),      # config 4
        ((1, 655360, 1024, 1, 1, 1e-3, 0), ROW),     # config 5
        ((1, T, 1056, 1, 1, 1e-3, 0), FAST), ((1, T, 2048, 1, 1, 1e-3, 0), FAST),   # beyond k_bpm_row's LDS band
        ((64, T, 256, 0, 1, 0.0, 0), FAST),          # eps = 0: k_bpm_row's default arithmetic needs a positive eps ...
        ((64, T, 256, 0, 1, 0.0, 2), ROW),           # ... its exact mode does not
        ((64, T, 512, 1, 1, 1e-3, 2), ROW), ((8, T, 256, 1, 1, 1e-3, 2), ROW),   # NR_FLAG_EXACT_GRADIENT: the same kernel
        ((64, T, 512, 1, 1, 1e-3, 8), FAST), ((64, T, 512, 1, 1, 1e-3, 8 | 2), FAST),  # NR_FLAG_K6_SCAN
        ((64, T, 512, 1, 1, 1e-3, 128), FAST), ((64, T, 512, 1, 1, 1e-3, 128 | 2), FAST),  # NR_FLAG_K6_LEGACY
        ((2, 40, 64, 1, 1, 1e-3, 65536), ROW), ((64, T, 512, 1, 1, 1e-3, 65536 | 2), ROW),  # NR_FLAG_K6_PX: ignored
    ]
    wrong = [(args, want, choice(*args)) for args, want in table if choice(*args) != want]
    assert not wrong, wrong


def test_argument_errors_do_not_need_a_gpu(lib_path):
    lib = _lib.load()
    # NULL pointers / bad sizes are rejected before any launch
    assert lib.nr_forward_face_index_map(None, None, None, None, None, None, 1, 1, 8, 0.1, 100.0, None, 0, None) == -1
    assert lib.nr_backward_depth_map(None, None, None, None, None, None, None, 1, 1, 8, None) == -1
    assert lib.nr_forward_texture_sampling(None, None, None, 1, None, None, None, None, None, None, 0, None,
                                           1, 1, 8, 2, 1e-3, 0, None) == -4
    # near <= 0 is accepted like in the reference (rasterize.py:331): the call gets as far as the workspace check
    assert lib.nr_forward_face_index_map(1, 1, None, None, None, None, 1, 1, 8, 0.0, 100.0, None, 0, None) == -3
    assert lib.nr_forward_face_index_map(1, 1, None, None, None, None, 1, 1, 8, -1.0, 100.0, None, 0, None) == -3
    # per-face light colours (nr_face_light): a descriptor without colours, and one whose texture_faces is neither F nor F / 2
    bad = _lib.FaceLight(None, 4, None, None)
    assert lib.nr_forward_rasterize_lit(bad, 1, None, 1, 1, 1, 1, 1, None, None, 1, 0, 1, 4, 8, 2, 0.1, 100.0, 1e-3, 0, None, 0,
                                        None) == -1
    bad = _lib.FaceLight(1, 3, None, None)
    assert lib.nr_forward_rasterize_lit(bad, 1, None, 1, 1, 1, 1, 1, None, None, 1, 0, 1, 4, 8, 2, 0.1, 100.0, 1e-3, 0, None, 0,
                                        None) == -2
    # ... the backward needs the cubes when the colours' gradient is asked for
    no_cubes = _lib.FaceLight(1, 4, None, 1)
    assert lib.nr_backward_rasterize_lit(no_cubes, 1, None, 1, 1, 1, 1, None, 1, None, None, 1, 1, 1, 4, 8, 2, 1e-3, 0, None, None,
                                         0, None) == -1


def test_cpu_tensors_are_refused():
    import torch
    import neural_renderer_amd as nr
    with pytest.raises(NotImplementedError):
        nr.rasterize_silhouettes(torch.rand(1, 2, 3, 3))
    with pytest.raises(Exception):
        nr.Rasterize(8, 0.1, 100, 1e-4, (0, 0, 0))  # nothing to draw (rasterize.py:25-27)
