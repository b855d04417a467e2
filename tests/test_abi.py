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
    return _build.build()


def test_header_and_binding_agree():
    names = declared_functions()
    assert 'nr_forward_face_index_map' in names and 'nr_backward_pixel_map' in names
    assert sorted(_lib.SIGNATURES) == names


def test_library_exports_every_declared_symbol(lib_path):
    lib = ctypes.CDLL(lib_path)
    for name in declared_functions():
        assert hasattr(lib, name), name


def test_host_only_entry_points(lib_path):
    lib = _lib.load()
    assert lib.nr_version() == 110
    assert lib.nr_error_string(0) == b'success'
    assert b'workspace' in lib.nr_error_string(-3)
    assert lib.nr_forward_workspace_bytes(64, 4928, 256) >= 64 * 4928 * (36 + 8)
    assert lib.nr_forward_workspace_bytes(0, 1, 1) == 0
    assert lib.nr_forward_workspace_bytes(1, 1, 20000) == 0


def test_argument_errors_do_not_need_a_gpu(lib_path):
    lib = _lib.load()
    # NULL pointers / bad sizes are rejected before any launch
    assert lib.nr_forward_face_index_map(None, None, None, None, None, 1, 1, 8, 0.1, 100.0, None, 0, None) == -1
    assert lib.nr_backward_depth_map(None, None, None, None, None, None, None, 1, 1, 8, None) == -1
    assert lib.nr_forward_texture_sampling(None, None, 1, None, None, None, None, None, None, 0, None,
                                           1, 1, 8, 2, 1e-3, 0, None) == -4


def test_cpu_tensors_are_refused():
    import torch
    import neural_renderer_amd as nr
    with pytest.raises(NotImplementedError):
        nr.rasterize_silhouettes(torch.rand(1, 2, 3, 3))
    with pytest.raises(Exception):
        nr.Rasterize(8, 0.1, 100, 1e-4, (0, 0, 0))  # nothing to draw (rasterize.py:25-27)
