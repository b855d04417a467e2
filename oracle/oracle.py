"""CPU ORACLE (test infrastructure, NOT product code) -- NumPy front-end of oracle/nr_oracle.c.

Restates, on the CPU, the orchestration of the reference's `Rasterize.forward_gpu/backward_gpu`
(neural_renderer/rasterize.py:467-513, :849-889), its public wrappers (`rasterize_rgbad`, :900-977)
and the small amount of Chainer glue (`Renderer`, `look_at`, `perspective`, `lighting`, `load_obj`,
`vertices_to_faces`, `get_points_from_angles`) that the reference's own fixtures are defined through.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The product package (neural_renderer_amd) never does.

Parity status: pinned against every fixture the reference ships for the path
(tests/test_oracle_golden.py); K7 backward_textures and K8 backward_depth_map are "parity unpinned"
(the reference has no effective test for them) and are pinned by the literal restatement + finite
differences only.
"""
import ctypes
import math
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SRC = os.path.join(_HERE, 'nr_oracle.c')
_LIB = os.path.join(_HERE, 'libnr_oracle.so')

DEFAULT_IMAGE_SIZE = 256            # rasterize.py:7
DEFAULT_ANTI_ALIASING = True        # rasterize.py:8
DEFAULT_NEAR = 0.1                  # rasterize.py:9
DEFAULT_FAR = 100                   # rasterize.py:10
DEFAULT_EPS = 1e-4                  # rasterize.py:11
DEFAULT_BACKGROUND_COLOR = (0, 0, 0)  # rasterize.py:12

_f32p = ctypes.POINTER(ctypes.c_float)
_i32p = ctypes.POINTER(ctypes.c_int32)
_lib = None


def build(force=False):
    """Compile nr_oracle.c with gcc (no contraction, no fast-math). Returns the .so path."""
    if force or (not os.path.exists(_LIB)) or os.path.getmtime(_LIB) < os.path.getmtime(_SRC):
        cmd = ['gcc', '-O2', '-std=c99', '-ffp-contract=off', '-fno-fast-math', '-fPIC', '-shared',
               '-fvisibility=hidden', _SRC, '-o', _LIB, '-lm']
        subprocess.check_call(cmd)
    return _LIB


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_LIB)
    return _lib


def _p(a, t):
    return None if a is None else a.ctypes.data_as(t)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


# ----------------------------------------------------------------------------------------------------
# Rasterize.forward_gpu / backward_gpu                                   (rasterize.py:467-513, 849-889)
class Rasterize(object):
    """Same constructor/call convention as the reference's chainer.Function (rasterize.py:19-64).

    `__call__(faces[, textures])` returns (rgb, alpha, depth) with None for disabled outputs and keeps
    every intermediate map on `self`; `backward(grad_rgb, grad_alpha, grad_depth)` returns
    (grad_faces,) or (grad_faces, grad_textures).
    """

    def __init__(self, image_size, near, far, eps, background_color, return_rgb=False, return_alpha=False,
                 return_depth=False, fix_batch_z=False):
        if not any((return_rgb, return_alpha, return_depth)):
            raise Exception  # rasterize.py:25-27
        self.image_size = int(image_size)
        self.near = float(near)
        self.far = float(far)
        self.eps = float(eps)
        self.background_color = background_color
        self.return_rgb = bool(return_rgb)
        self.return_alpha = bool(return_alpha)
        self.return_depth = bool(return_depth)
        self.fix_batch_z = bool(fix_batch_z)
        self.visits = None

    def __call__(self, faces, textures=None):
        L = lib()
        self.faces = _f32(faces).copy()                      # :470
        assert self.faces.ndim == 4 and self.faces.shape[2:] == (3, 3)
        bs, nf = self.faces.shape[:2]
        s = self.image_size
        self.batch_size, self.num_faces = bs, nf
        if self.return_rgb:
            self.textures = _f32(textures)
            assert self.textures.ndim == 6 and self.textures.shape[:2] == (bs, nf)
            ts = self.textures.shape[2]
            assert ts >= 2 and self.textures.shape[2:] == (ts, ts, ts, 3)
            self.texture_size = ts

        # :478-496
        self.face_index_map = np.full((bs, s, s), -1, np.int32)
        self.weight_map = np.zeros((bs, s, s, 3), np.float32)
        self.depth_map = np.zeros((bs, s, s), np.float32) + np.float32(self.far)
        self.rgb_map = np.zeros((bs, s, s, 3), np.float32) if self.return_rgb else None
        self.sampling_index_map = np.zeros((bs, s, s, 8), np.int32) if self.return_rgb else None
        self.sampling_weight_map = np.zeros((bs, s, s, 8), np.float32) if self.return_rgb else None
        self.alpha_map = np.zeros((bs, s, s), np.float32) if self.return_alpha else None
        self.face_inv_map = np.zeros((bs, s, s, 3, 3), np.float32) if self.return_depth else None

        # :499 forward_face_index_map_gpu (safe path: K1 then K2)
        self.faces_inv = np.zeros_like(self.faces)
        L.oracle_forward_face_inv(_p(self.faces, _f32p), _p(self.faces_inv, _f32p), bs, nf, s)
        L.oracle_forward_face_index_map(
            _p(self.faces, _f32p), _p(self.faces_inv, _f32p), _p(self.face_index_map, _i32p),
            _p(self.weight_map, _f32p), _p(self.depth_map, _f32p), _p(self.face_inv_map, _f32p),
            bs, nf, s, ctypes.c_double(self.near), ctypes.c_double(self.far), int(self.return_depth))
        # :500 forward_texture_sampling
        if self.return_rgb:
            L.oracle_forward_texture_sampling(
                _p(self.faces, _f32p), _p(self.textures, _f32p), _p(self.face_index_map, _i32p),
                _p(self.weight_map, _f32p), _p(self.depth_map, _f32p), _p(self.rgb_map, _f32p),
                _p(self.sampling_index_map, _i32p), _p(self.sampling_weight_map, _f32p),
                bs, nf, s, self.texture_size, ctypes.c_double(self.eps), int(self.fix_batch_z))
        # :501-502 forward_background_gpu, forward_alpha_map_gpu
        bg = None
        per_batch = 0
        if self.return_rgb:
            bg = _f32(self.background_color)
            assert bg.shape in ((3,), (bs, 3))
            per_batch = int(bg.ndim == 2)
        L.oracle_forward_background_alpha(
            _p(self.face_index_map, _i32p), _p(self.rgb_map, _f32p), _p(self.alpha_map, _f32p),
            _p(bg, _f32p), per_batch, bs, s)

        # :505-513
        rgb_r = self.rgb_map if self.return_rgb else None
        alpha_r = self.alpha_map.copy() if self.return_alpha else None
        depth_r = self.depth_map.copy() if self.return_depth else None
        return rgb_r, alpha_r, depth_r

    def backward(self, grad_rgb=None, grad_alpha=None, grad_depth=None, accumulate_double=False):
        """accumulate_double: keep K6's running sums in double (NOT the reference; see nr_oracle.c)."""
        L = lib()
        bs, nf, s = self.batch_size, self.num_faces, self.image_size
        # :851-855
        self.grad_faces = np.zeros_like(self.faces)
        self.grad_textures = np.zeros_like(self.textures) if self.return_rgb else None
        # :858-878 (None -> zeros)
        g_rgb = g_alpha = g_depth = None
        if self.return_rgb:
            g_rgb = _f32(grad_rgb) if grad_rgb is not None else np.zeros_like(self.rgb_map)
            assert g_rgb.shape == self.rgb_map.shape
        if self.return_alpha:
            g_alpha = _f32(grad_alpha) if grad_alpha is not None else np.zeros_like(self.alpha_map)
            assert g_alpha.shape == self.alpha_map.shape
        if self.return_depth:
            g_depth = _f32(grad_depth) if grad_depth is not None else np.zeros_like(self.depth_map)
            assert g_depth.shape == self.depth_map.shape

        # :881 backward_pixel_map_gpu
        visits = ctypes.c_longlong(0)
        L.oracle_backward_pixel_map(
            _p(self.faces, _f32p), _p(self.face_index_map, _i32p), _p(self.rgb_map, _f32p),
            _p(self.alpha_map, _f32p), _p(g_rgb, _f32p), _p(g_alpha, _f32p), _p(self.grad_faces, _f32p),
            bs, nf, s, ctypes.c_double(self.eps), int(self.return_rgb), int(self.return_alpha),
            ctypes.byref(visits), int(accumulate_double))
        self.visits = visits.value
        _f64p = ctypes.POINTER(ctypes.c_double)
        # :882 backward_textures_gpu
        if self.return_rgb:
            acc = np.zeros(self.grad_textures.shape, np.float64) if accumulate_double else None
            L.oracle_backward_textures(
                _p(self.face_index_map, _i32p), _p(self.sampling_weight_map, _f32p),
                _p(self.sampling_index_map, _i32p), _p(g_rgb, _f32p), _p(self.grad_textures, _f32p),
                bs, nf, s, self.texture_size, _p(acc, _f64p))
            if acc is not None:
                self.grad_textures = acc.astype(np.float32)
        # :883 backward_depth_map_gpu
        if self.return_depth:
            acc = self.grad_faces.astype(np.float64) if accumulate_double else None
            L.oracle_backward_depth_map(
                _p(self.faces, _f32p), _p(self.depth_map, _f32p), _p(self.face_index_map, _i32p),
                _p(self.face_inv_map, _f32p), _p(self.weight_map, _f32p), _p(g_depth, _f32p),
                _p(self.grad_faces, _f32p), bs, nf, s, _p(acc, _f64p))
            if acc is not None:
                self.grad_faces = acc.astype(np.float32)
        if self.return_rgb:
            return self.grad_faces, self.grad_textures
        return self.grad_faces,


# ----------------------------------------------------------------------------------------------------
# public wrappers                                                                (rasterize.py:900-1060)
def _avg_pool2(x):
    """cf.average_pooling_2d(x, 2, 2) on the last two axes (rasterize.py:965-969)."""
    return (x[..., 0::2, 0::2] + x[..., 0::2, 1::2] + x[..., 1::2, 0::2] + x[..., 1::2, 1::2]) * np.float32(0.25)


def rasterize_rgbad(faces, textures=None, image_size=DEFAULT_IMAGE_SIZE, anti_aliasing=DEFAULT_ANTI_ALIASING,
                    near=DEFAULT_NEAR, far=DEFAULT_FAR, eps=DEFAULT_EPS,
                    background_color=DEFAULT_BACKGROUND_COLOR, return_rgb=True, return_alpha=True,
                    return_depth=True, fix_batch_z=False, return_function=False):
    s = image_size * 2 if anti_aliasing else image_size   # :945-951
    fn = Rasterize(s, near, far, eps, background_color, return_rgb, return_alpha, return_depth, fix_batch_z)
    rgb, alpha, depth = fn(faces, textures) if textures is not None else fn(faces)
    if return_rgb:                                        # :954-960
        rgb = rgb.transpose((0, 3, 1, 2))[:, :, ::-1, :]
    if return_alpha:
        alpha = alpha[:, ::-1, :]
    if return_depth:
        depth = depth[:, ::-1, :]
    if anti_aliasing:                                     # :962-969
        if return_rgb:
            rgb = _avg_pool2(rgb)
        if return_alpha:
            alpha = _avg_pool2(alpha)
        if return_depth:
            depth = _avg_pool2(depth)
    ret = {'rgb': rgb if return_rgb else None, 'alpha': alpha if return_alpha else None,
           'depth': depth if return_depth else None}
    if return_function:
        ret['function'] = fn
    return ret


def rgbad_backward(fn, anti_aliasing, grad_rgb=None, grad_alpha=None, grad_depth=None):
    """Back-propagate image-space gradients (API layout: rgb [B,3,is,is], alpha/depth [B,is,is]) through
    the average pooling, flip and transpose of `rasterize_rgbad`, then through `fn.backward`."""
    def up(g):
        if g is None:
            return None
        g = _f32(g)
        if anti_aliasing:
            g = np.repeat(np.repeat(g, 2, axis=-2), 2, axis=-1) * np.float32(0.25)
        return g
    g_rgb, g_alpha, g_depth = up(grad_rgb), up(grad_alpha), up(grad_depth)
    if g_rgb is not None:
        g_rgb = np.ascontiguousarray(g_rgb[:, :, ::-1, :].transpose((0, 2, 3, 1)))
    if g_alpha is not None:
        g_alpha = np.ascontiguousarray(g_alpha[:, ::-1, :])
    if g_depth is not None:
        g_depth = np.ascontiguousarray(g_depth[:, ::-1, :])
    return fn.backward(g_rgb, g_alpha, g_depth)


def rasterize(faces, textures, image_size=DEFAULT_IMAGE_SIZE, anti_aliasing=DEFAULT_ANTI_ALIASING,
              near=DEFAULT_NEAR, far=DEFAULT_FAR, eps=DEFAULT_EPS, background_color=DEFAULT_BACKGROUND_COLOR):
    return rasterize_rgbad(faces, textures, image_size, anti_aliasing, near, far, eps, background_color,
                           True, False, False)['rgb']    # :1007-1008


def rasterize_silhouettes(faces, image_size=DEFAULT_IMAGE_SIZE, anti_aliasing=DEFAULT_ANTI_ALIASING,
                          near=DEFAULT_NEAR, far=DEFAULT_FAR, eps=DEFAULT_EPS):
    return rasterize_rgbad(faces, None, image_size, anti_aliasing, near, far, eps, None,
                           False, True, False)['alpha']  # :1034


def rasterize_depth(faces, image_size=DEFAULT_IMAGE_SIZE, anti_aliasing=DEFAULT_ANTI_ALIASING,
                    near=DEFAULT_NEAR, far=DEFAULT_FAR, eps=DEFAULT_EPS):
    return rasterize_rgbad(faces, None, image_size, anti_aliasing, near, far, eps, None,
                           False, False, True)['depth']  # :1060


# ----------------------------------------------------------------------------------------------------
# glue (float32 NumPy restatements of the Chainer graph around the rasterizer)
def load_obj(filename_obj, normalization=True):
    """neural_renderer/load_obj.py:147-197 (vertices + fan-triangulated faces; no textures)."""
    vertices, faces = [], []
    with open(filename_obj) as f:
        for line in f:
            t = line.split()
            if not t:
                continue
            if t[0] == 'v':
                vertices.append([float(v) for v in t[1:4]])
            elif t[0] == 'f':
                vs = [int(s.split('/')[0]) for s in t[1:]]
                for i in range(len(vs) - 2):
                    faces.append((vs[0], vs[i + 1], vs[i + 2]))
    vertices = np.vstack(vertices).astype('float32')
    faces = np.vstack(faces).astype('int32') - 1
    if normalization:
        vertices = normalize_vertices(vertices)
    return vertices, faces


def normalize_vertices(vertices):
    """load_obj.py:188-192 (in-place float32 arithmetic)."""
    vertices = np.array(vertices, np.float32)
    vertices -= vertices.min(0)[None, :]
    vertices /= np.abs(vertices).max()
    vertices *= 2
    vertices -= vertices.max(0)[None, :] / 2
    return vertices


def get_points_from_angles(distance, elevation, azimuth, degrees=True):
    """get_points_from_angles.py:6-13 (scalar form)."""
    if degrees:
        elevation = math.radians(elevation)
        azimuth = math.radians(azimuth)
    return (distance * math.cos(elevation) * math.sin(azimuth),
            distance * math.sin(elevation),
            -distance * math.cos(elevation) * math.cos(azimuth))


def _normalize(x, eps=1e-5):
    """chainer.functions.normalize: x / (||x||_2 + eps) along axis 1 (look_at.py:30-32)."""
    norm = np.sqrt(np.sum(x * x, axis=1, keepdims=True, dtype=np.float32)).astype(np.float32) + np.float32(eps)
    return (x / norm).astype(np.float32)


def look_at(vertices, eye, at=None, up=None):
    """look_at.py:7-46."""
    vertices = _f32(vertices)
    bs = vertices.shape[0]
    at = np.array([0, 0, 0], 'float32') if at is None else _f32(at)
    up = np.array([0, 1, 0], 'float32') if up is None else _f32(up)
    eye = _f32(eye)
    if eye.ndim == 1:
        eye = np.tile(eye[None, :], (bs, 1))
    if at.ndim == 1:
        at = np.tile(at[None, :], (bs, 1))
    if up.ndim == 1:
        up = np.tile(up[None, :], (bs, 1))
    z_axis = _normalize(at - eye)
    x_axis = _normalize(np.cross(up, z_axis).astype(np.float32))
    y_axis = _normalize(np.cross(z_axis, x_axis).astype(np.float32))
    r = np.concatenate((x_axis[:, None, :], y_axis[:, None, :], z_axis[:, None, :]), axis=1)
    vertices = vertices - eye[:, None, :]
    return np.matmul(vertices, r.transpose(0, 2, 1)).astype(np.float32)


def perspective(vertices, angle=30.):
    """perspective.py:5-19 (note pi = 3.1416, :10)."""
    vertices = _f32(vertices)
    angle = np.float32(angle) / np.float32(180.) * np.float32(3.1416)
    width = np.tan(angle, dtype=np.float32)
    z = vertices[:, :, 2]
    x = vertices[:, :, 0] / z / width
    y = vertices[:, :, 1] / z / width
    return np.stack((x, y, z), axis=2).astype(np.float32)


def vertices_to_faces(vertices, faces):
    """vertices_to_faces.py:4-21."""
    vertices = _f32(vertices)
    bs, nv = vertices.shape[:2]
    faces = faces + (np.arange(bs, dtype='int32') * nv)[:, None, None]
    return vertices.reshape((bs * nv, 3))[faces]


def lighting(faces, textures, intensity_ambient=0.5, intensity_directional=0.5, color_ambient=(1, 1, 1),
             color_directional=(1, 1, 1), direction=(0, 1, 0)):
    """lighting.py:8-51."""
    faces = _f32(faces)
    textures = _f32(textures)
    bs, nf = faces.shape[:2]
    color_ambient = np.broadcast_to(_f32(color_ambient).reshape(-1, 3), (bs, 3))
    color_directional = np.broadcast_to(_f32(color_directional).reshape(-1, 3), (bs, 3))
    direction = np.broadcast_to(_f32(direction).reshape(-1, 3), (bs, 3))
    light = np.zeros((bs, nf, 3), 'float32')
    if intensity_ambient != 0:
        light = light + np.float32(intensity_ambient) * color_ambient[:, None, :]
    if intensity_directional != 0:
        f = faces.reshape((bs * nf, 3, 3))
        v10 = f[:, 0] - f[:, 1]
        v12 = f[:, 2] - f[:, 1]
        normals = _normalize(np.cross(v10, v12).astype(np.float32)).reshape((bs, nf, 3))
        cos = np.maximum(np.sum(normals * direction[:, None, :], axis=2, dtype=np.float32), np.float32(0))
        light = light + np.float32(intensity_directional) * (color_directional[:, None, :] * cos[:, :, None])
    return (textures * light[:, :, None, None, None, :]).astype(np.float32)


class Renderer(object):
    """renderer.py:8-107 on NumPy arrays."""

    def __init__(self):
        self.image_size = 256
        self.anti_aliasing = True
        self.background_color = [0, 0, 0]
        self.fill_back = True
        self.perspective = True
        self.viewing_angle = 30
        self.eye = [0, 0, -(1. / math.tan(math.radians(self.viewing_angle)) + 1)]
        self.camera_mode = 'look_at'
        self.camera_direction = [0, 0, 1]
        self.near = 0.1
        self.far = 100
        self.light_intensity_ambient = 0.5
        self.light_intensity_directional = 0.5
        self.light_color_ambient = [1, 1, 1]
        self.light_color_directional = [1, 1, 1]
        self.light_direction = [0, 1, 0]
        self.rasterizer_eps = 1e-3

    def _camera(self, vertices):
        if self.camera_mode == 'look_at':
            vertices = look_at(vertices, self.eye)
        if self.perspective:
            vertices = perspective(vertices, angle=self.viewing_angle)
        return vertices

    def project(self, vertices, faces):
        """fill_back + camera + vertices_to_faces: the `faces` tensor handed to rasterize* (renderer.py:37-51)."""
        if self.fill_back:
            faces = np.concatenate((faces, faces[:, :, ::-1]), axis=1)
        return vertices_to_faces(self._camera(vertices), faces)

    def render_silhouettes(self, vertices, faces):
        return rasterize_silhouettes(self.project(vertices, faces), self.image_size, self.anti_aliasing)

    def render_depth(self, vertices, faces):
        return rasterize_depth(self.project(vertices, faces), self.image_size, self.anti_aliasing)

    def render(self, vertices, faces, textures):
        if self.fill_back:
            faces = np.concatenate((faces, faces[:, :, ::-1]), axis=1)
            textures = np.concatenate((textures, textures.transpose((0, 1, 4, 3, 2, 5))), axis=1)
        faces_lighting = vertices_to_faces(vertices, faces)
        textures = lighting(faces_lighting, textures, self.light_intensity_ambient,
                            self.light_intensity_directional, self.light_color_ambient,
                            self.light_color_directional, self.light_direction)
        faces = vertices_to_faces(self._camera(vertices), faces)
        return rasterize(faces, textures, self.image_size, self.anti_aliasing, self.near, self.far,
                         self.rasterizer_eps, self.background_color)
