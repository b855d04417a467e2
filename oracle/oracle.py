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
        cmd = ['gcc', '-O2', '-std=c99', '-fopenmp', '-ffp-contract=off', '-fno-fast-math', '-fPIC', '-shared',
               '-fvisibility=hidden', _SRC, '-o', _LIB, '-lm']
        subprocess.check_call(cmd)
    return _LIB


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_LIB)
        if _lib.oracle_version() < 4:  # a stale prebuilt library
            build(force=True)
            _lib = ctypes.CDLL(_LIB)
        env = os.environ.get('NR_ORACLE_THREADS')
        if env:
            _lib.oracle_set_threads(int(env))
    return _lib


def set_threads(n):
    """Number of OpenMP threads of the C oracle (0 = all cores, the default).  Results do not depend on it."""
    lib().oracle_set_threads(int(n))


def get_threads():
    return int(lib().oracle_get_threads())


def set_contraction(on):
    """Contraction study only (nr_oracle.c: oracle_set_contraction): K1 / K2 with the `a*b+c` of rasterize.py:258, :261-269,
    :317-319 evaluated as fused multiply-adds, the way nvcc is allowed to compile the reference.  The parity convention is
    the un-fused reading (off, the default); callers must switch it off again."""
    lib().oracle_set_contraction(1 if on else 0)


def get_contraction():
    return bool(lib().oracle_get_contraction())


# Above this many (pixel, face) pairs the forward uses the cache-blocked evaluation order of K2 (bit-identical,
# tests/test_oracle_threads.py); `Rasterize.blocked` overrides.
BLOCKED_K2_THRESHOLD = 2 ** 33


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
        self.blocked = None  # None: by size; True / False: force the cache-blocked / literal K2 loop order
        # True: visibility by the reference's "unsafe" kernel K3 (rasterize.py:102-236, USE_UNSAFE_IMPLEMENTATION) in its race-free
        # sequential emulation instead of K1 + K2 (nr_oracle.c: oracle_forward_face_index_map_unsafe)
        self.unsafe = False

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

        # :499 forward_face_index_map_gpu (safe path: K1 then K2; `unsafe`: K3, :102-236)
        self.faces_inv = np.zeros_like(self.faces)
        if self.unsafe:
            L.oracle_forward_face_index_map_unsafe(
                _p(self.faces, _f32p), _p(self.face_index_map, _i32p), _p(self.weight_map, _f32p), _p(self.depth_map, _f32p),
                _p(self.face_inv_map, _f32p), bs, nf, s, ctypes.c_double(self.near), ctypes.c_double(self.far),
                int(self.return_depth))
        else:
            L.oracle_forward_face_inv(_p(self.faces, _f32p), _p(self.faces_inv, _f32p), bs, nf, s)
            blocked = self.blocked if self.blocked is not None else bs * s * s * nf >= BLOCKED_K2_THRESHOLD
            k2 = L.oracle_forward_face_index_map_blocked if blocked else L.oracle_forward_face_index_map
            k2(
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

    def backward(self, grad_rgb=None, grad_alpha=None, grad_depth=None, accumulate_double=False, skip_textures=False):
        """accumulate_double: keep K6's running sums in double (NOT the reference; see nr_oracle.c).
        skip_textures: leave K7 out (its output stays zero) -- for callers that only want grad_faces of a huge mesh."""
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
        if self.return_rgb and not skip_textures:
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
def _read_image(path):
    """skimage.io.imread(path) as the reference uses it (load_obj.py:83): uint8 [H,W,3]."""
    from PIL import Image
    return np.asarray(Image.open(path).convert('RGB'))


def load_mtl(filename_mtl):
    """load_obj.py:9-22: diffuse colour (Kd) and texture file (map_Kd) per material."""
    texture_filenames, colors = {}, {}
    material_name = ''
    with open(filename_mtl) as f:
        for line in f:
            t = line.split()
            if not t:
                continue
            if t[0] == 'newmtl':
                material_name = t[1]
            if t[0] == 'map_Kd':
                texture_filenames[material_name] = t[1]
            if t[0] == 'Kd':
                colors[material_name] = np.array([float(x) for x in t[1:4]])
    return colors, texture_filenames


def bake_texture_image(image, faces_uv, is_update, textures):
    """K10, load_obj.py:87-144: for every texel (i0,i1,i2) of every face with is_update != 0, the barycentric point
    dim/sum(dim) of the face's uv triangle is looked up in `image` ([H,W,3] float32, ALREADY flipped vertically, :85) with
    bilinear filtering.  In place on textures [Nf,ts,ts,ts,3].  Reads that the reference performs outside the image
    (uv exactly 1, negative uv: undefined behaviour there) are clamped to the nearest valid flat index -- they carry a zero
    weight whenever the reference's result is defined."""
    nf, ts = textures.shape[:2]
    h, w = image.shape[:2]
    img = np.ascontiguousarray(image, np.float32).reshape(-1, 3)
    idx = np.arange(ts, dtype=np.int64)
    grid = (idx.astype(np.float64) / (ts - 1.)).astype(np.float32)                     # :98-100 (int / double -> float)
    d0, d1, d2 = np.meshgrid(grid, grid, grid, indexing='ij')
    with np.errstate(all='ignore'):
        total = (d0 + d1) + d2                                                         # :103
        d0, d1, d2 = d0 / total, d1 / total, d2 / total                                # :104-106 (0/0 = NaN at texel 0,0,0)
        for fn in np.nonzero(np.asarray(is_update) != 0)[0]:                           # :110
            f = faces_uv[fn]
            pos_x = ((f[0, 0] * d0 + f[1, 0] * d1) + f[2, 0] * d2) * np.float32(w - 1)  # :112-113
            pos_y = ((f[0, 1] * d0 + f[1, 1] * d1) + f[2, 1] * d2) * np.float32(h - 1)  # :114-115
            xi, yi = _f2i_arr(pos_x), _f2i_arr(pos_y)
            yi1 = _f2i_arr(pos_y + np.float32(1))
            wx1 = pos_x - xi.astype(np.float32)                                        # :118-121
            wx0 = np.float32(1) - wx1
            wy1 = pos_y - yi.astype(np.float32)
            wy0 = np.float32(1) - wy1

            def px(row, col):
                return img[np.clip(row * w + col, 0, h * w - 1)]
            c = np.zeros(pos_x.shape + (3,), np.float32)                               # :123-128
            c = c + px(yi, xi) * (wx0 * wy0)[..., None]
            c = c + px(yi1, xi) * (wx0 * wy1)[..., None]
            c = c + px(yi, xi + 1) * (wx1 * wy0)[..., None]
            c = c + px(yi1, xi + 1) * (wx1 * wy1)[..., None]
            textures[fn] = c
    return textures


def _f2i_arr(x):
    """CUDA (int)x on an array: truncate, saturate, NaN -> 0."""
    x = np.asarray(x, np.float64)
    return np.where(np.isnan(x), 0.0, np.clip(x, -2147483648.0, 2147483647.0)).astype(np.int64)


def parse_obj_texture_faces(filename_obj):
    """load_obj.py:26-62: uv coordinates per face corner and the material name in force at each face."""
    uv = []
    faces, material_names = [], []
    material_name = ''
    with open(filename_obj) as f:
        lines = f.readlines()
    for line in lines:
        t = line.split()
        if t and t[0] == 'vt':
            uv.append([float(v) for v in t[1:3]])
    uv = np.vstack(uv).astype('float32')
    for line in lines:
        t = line.split()
        if not t:
            continue
        if t[0] == 'f':
            vs = t[1:]
            ids = [int(v.split('/')[1]) if '/' in v else 0 for v in vs]                # :44-56
            for i in range(len(vs) - 2):
                faces.append((ids[0], ids[i + 1], ids[i + 2]))
                material_names.append(material_name)
        if t[0] == 'usemtl':
            material_name = t[1]
    faces = np.vstack(faces).astype('int32') - 1
    faces_uv = uv[faces]                                                               # :62 (index -1 wraps like NumPy)
    faces_uv[1 < faces_uv] = faces_uv[1 < faces_uv] % 1                                # :64
    return faces_uv, material_names


def load_textures(filename_obj, filename_mtl, texture_size):
    """load_obj.py:25-144."""
    faces_uv, material_names = parse_obj_texture_faces(filename_obj)
    colors, texture_filenames = load_mtl(filename_mtl)
    textures = np.zeros((faces_uv.shape[0], texture_size, texture_size, texture_size, 3), 'float32') + 0.5   # :69
    names = np.array(material_names)
    for material_name, color in colors.items():                                        # :73-77
        textures[names == material_name] = color.astype(np.float32)[None, None, None, None, :]
    for material_name, filename_texture in texture_filenames.items():                  # :80-143
        path = os.path.join(os.path.dirname(filename_obj), filename_texture)
        image = _read_image(path).astype('float32') / 255.                             # :83
        image = image[::-1, ::1]                                                       # :85
        bake_texture_image(image, faces_uv, (names == material_name).astype('int32'), textures)
    return textures


def load_obj(filename_obj, normalization=True, texture_size=4, load_texture=False):
    """neural_renderer/load_obj.py:147-197 (vertices + fan-triangulated faces [+ baked textures])."""
    vertices, faces = [], []
    with open(filename_obj) as f:
        lines = f.readlines()
    for line in lines:
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
    textures = None
    if load_texture:                                                                   # :177-185
        for line in lines:
            if line.startswith('mtllib'):
                filename_mtl = os.path.join(os.path.dirname(filename_obj), line.split()[1])
                textures = load_textures(filename_obj, filename_mtl, texture_size)
        if textures is None:
            raise Exception('Failed to load textures.')
    if normalization:
        vertices = normalize_vertices(vertices)
    if load_texture:
        return vertices, faces, textures
    return vertices, faces


def create_texture_image(textures, texture_size_out=16):
    """save_obj.py:10-147 (K11): atlas image [tile_h*tso, tile_w*tso, 3] (flipped vertically, :143) and the per-face uv
    triangles [Nf,3,2] normalised to [0,1] (:140-141).  Tiles beyond the last face stay 0."""
    textures = _f32(textures)
    num_faces, tsi = textures.shape[:2]
    tso = texture_size_out
    tile_width = int((num_faces - 1.) ** 0.5) + 1                                      # :12
    tile_height = int((num_faces - 1.) / tile_width) + 1                               # :13
    height, width = tile_height * tso, tile_width * tso
    image = np.zeros((height, width, 3), 'float32')
    vertices = np.zeros((num_faces, 3, 2), 'float32')                                  # :16-25
    face_nums = np.arange(num_faces)
    column = face_nums % tile_width
    row = face_nums // tile_width
    vertices[:, 0, 0] = column * tso
    vertices[:, 0, 1] = row * tso
    vertices[:, 1, 0] = column * tso
    vertices[:, 1, 1] = (row + 1) * tso - 1
    vertices[:, 2, 0] = (column + 1) * tso - 1
    vertices[:, 2, 1] = (row + 1) * tso - 1

    tex = textures.reshape(num_faces, tsi * tsi * tsi, 3)
    ys, xs = np.meshgrid(np.arange(height), np.arange(width), indexing='ij')
    fn = xs // tso + (ys // tso) * tile_width                                          # :39-41
    valid = fn < num_faces
    fnc = np.minimum(fn, num_faces - 1)
    p0, p1, p2 = vertices[fnc, 0], vertices[fnc, 1], vertices[fnc, 2]
    xf, yf = xs.astype(np.float32), ys.astype(np.float32)
    with np.errstate(all='ignore'):
        inv = [p1[..., 1] - p2[..., 1], p2[..., 0] - p1[..., 0], p1[..., 0] * p2[..., 1] - p2[..., 0] * p1[..., 1],   # :54-57
               p2[..., 1] - p0[..., 1], p0[..., 0] - p2[..., 0], p2[..., 0] * p0[..., 1] - p0[..., 0] * p2[..., 1],
               p0[..., 1] - p1[..., 1], p1[..., 0] - p0[..., 0], p0[..., 0] * p1[..., 1] - p1[..., 0] * p0[..., 1]]
        den = (p2[..., 0] * (p0[..., 1] - p1[..., 1]) + p0[..., 0] * (p1[..., 1] - p2[..., 1])
               + p1[..., 0] * (p2[..., 1] - p0[..., 1]))                                 # :58-61
        inv = [m / den for m in inv]
        weight = [inv[3 * k] * xf + inv[3 * k + 1] * yf + inv[3 * k + 2] for k in range(3)]   # :67
        weight_sum = ((np.float32(0) + weight[0]) + weight[1]) + weight[2]
        weight = [(w.astype(np.float64) / (weight_sum.astype(np.float64) + 1e-5)).astype(np.float32) for w in weight]  # :70
        tif = []
        for k in range(3):                                                             # :73-79
            t = weight[k] * np.float32(tsi - 1)
            t = np.fmax(t.astype(np.float64), 0.).astype(np.float32)
            t = np.fmin(t.astype(np.float64), (tsi - 1) - 1e-5).astype(np.float32)
            tif.append(t)
        ti = [_f2i_arr(t) for t in tif]
        frac = [t - i.astype(np.float32) for t, i in zip(tif, ti)]
        pixel = np.zeros((height, width, 3), np.float32)
        for pn in range(8):                                                            # :82-97
            w = np.ones((height, width), np.float32)
            idx = []
            for k in range(3):
                if (pn >> k) % 2 == 0:
                    w = w * (np.float32(1) - frac[k])
                    idx.append(ti[k])
                else:
                    w = w * frac[k]
                    idx.append(ti[k] + 1)
            isc = idx[0] * tsi * tsi + idx[1] * tsi + idx[2]
            pixel = pixel + w[..., None] * tex[fnc, np.clip(isc, 0, tsi ** 3 - 1)]
    image[valid] = pixel[valid]
    seam = (ys % tso + 1) == (xs % tso)                                                # :129-132
    src = image[ys, np.maximum(xs - 1, 0)]
    image[seam] = src[seam]
    vertices[:, :, 0] /= (image.shape[1] - 1)                                          # :140-141
    vertices[:, :, 1] /= (image.shape[0] - 1)
    return image[::-1, ::1], vertices


def toimage_bytes(image, cmin=0.0, cmax=1.0):
    """scipy.misc.toimage(image, cmin, cmax) byte scaling (save_obj.py:158): (x - cmin) * 255 / (cmax - cmin), clip, + 0.5."""
    data = (np.asarray(image, np.float64) - cmin) * (255.0 / (cmax - cmin))
    return (data.clip(0, 255) + 0.5).astype(np.uint8)


def save_obj(filename, vertices, faces, textures=None):
    """save_obj.py:150-191, byte for byte the reference's text format."""
    from PIL import Image
    assert vertices.ndim == 2
    assert faces.ndim == 2
    if textures is not None:
        filename_mtl = filename[:-4] + '.mtl'
        filename_texture = filename[:-4] + '.png'
        material_name = 'material_1'
        texture_image, vertices_textures = create_texture_image(textures)
        Image.fromarray(toimage_bytes(texture_image)).save(filename_texture)
    with open(filename, 'w') as f:
        f.write('# %s\n' % os.path.basename(filename))
        f.write('#\n')
        f.write('\n')
        if textures is not None:
            f.write('mtllib %s\n\n' % os.path.basename(filename_mtl))
        for vertex in vertices:
            f.write('v %.8f %.8f %.8f\n' % (vertex[0], vertex[1], vertex[2]))
        f.write('\n')
        if textures is not None:
            for vertex in vertices_textures.reshape((-1, 2)):
                f.write('vt %.8f %.8f\n' % (vertex[0], vertex[1]))
            f.write('\n')
            f.write('usemtl %s\n' % material_name)
            for i, face in enumerate(faces):
                f.write('f %d/%d %d/%d %d/%d\n' % (
                    face[0] + 1, 3 * i + 1, face[1] + 1, 3 * i + 2, face[2] + 1, 3 * i + 3))
            f.write('\n')
        else:
            for face in faces:
                f.write('f %d %d %d\n' % (face[0] + 1, face[1] + 1, face[2] + 1))
    if textures is not None:
        with open(filename_mtl, 'w') as f:
            f.write('newmtl %s\n' % material_name)
            f.write('map_Kd %s\n' % os.path.basename(filename_texture))


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


def project_backward(vertices, faces_idx, eye, grad_faces, angle=30., fill_back=True):
    """Backward of Renderer.project (fill_back + look_at + perspective + vertices_to_faces) w.r.t. the vertices, in float64:
    grad_faces [B, F', 3, 3] -> grad_vertices [B, Nv, 3].  The chain rule of look_at.py:42-44 (v' = R (v - eye)),
    perspective.py:15-17 (x / z / w, y / z / w, z) and the get_item backward (scatter-add) of vertices_to_faces.py:19-21."""
    vertices = np.asarray(vertices, np.float64)
    bs, nv = vertices.shape[:2]
    eye = np.broadcast_to(np.asarray(eye, np.float64).reshape(-1, 3), (bs, 3))
    z_axis = np.float64(1) * (0 - eye)
    z_axis = z_axis / (np.linalg.norm(z_axis, axis=1, keepdims=True) + 1e-5)
    x_axis = np.cross(np.broadcast_to([0., 1., 0.], (bs, 3)), z_axis)
    x_axis = x_axis / (np.linalg.norm(x_axis, axis=1, keepdims=True) + 1e-5)
    y_axis = np.cross(z_axis, x_axis)
    y_axis = y_axis / (np.linalg.norm(y_axis, axis=1, keepdims=True) + 1e-5)
    r = np.stack((x_axis, y_axis, z_axis), axis=1)                              # [B,3,3], rows = axes
    cam = np.matmul(vertices - eye[:, None, :], r.transpose(0, 2, 1))           # camera-space points
    width = math.tan(float(np.float32(angle) / np.float32(180.) * np.float32(3.1416)))
    idx = np.concatenate((faces_idx, faces_idx[:, :, ::-1]), axis=1) if fill_back else faces_idx
    g_proj = np.zeros((bs, nv, 3))
    for b in range(bs):                                                         # scatter-add of the gather
        np.add.at(g_proj[b], idx[b].reshape(-1), np.asarray(grad_faces[b], np.float64).reshape(-1, 3))
    zw = cam[..., 2] * width
    g_cam = np.stack((g_proj[..., 0] / zw, g_proj[..., 1] / zw,
                      g_proj[..., 2] - (g_proj[..., 0] * cam[..., 0] + g_proj[..., 1] * cam[..., 1]) / (cam[..., 2] * zw)), axis=2)
    return np.matmul(g_cam, r)                                                  # g_world = R^T g_cam


# ----------------------------------------------------------------------------------------------------
# optimizer                                                                  (optimizers.py:9-39)
def adam_update(param, grad, m, v, lr, one_minus_beta1, one_minus_beta2, eps):
    """optimizers.py:22-34 in float32, in place: elements with grad == 0 keep parameter and moments."""
    lr, b1, b2, eps = np.float32(lr), np.float32(one_minus_beta1), np.float32(one_minus_beta2), np.float32(eps)
    on = grad != 0
    g = grad[on]
    mi = m[on] + b1 * (g - m[on])
    vi = v[on] + b2 * (g * g - v[on])
    vi = np.where(vi < 0, np.float32(0), vi)
    m[on], v[on] = mi, vi
    param[on] = param[on] - lr * mi / (np.sqrt(vi) + eps)


class Adam(object):
    """chainer.optimizers.Adam with the reference's masked rule; `lr_mult` = the parameter's `.lr` (optimizers.py:20)."""

    def __init__(self, alpha=0.001, beta1=0.9, beta2=0.999, eps=1e-8):
        self.alpha, self.beta1, self.beta2, self.eps, self.t = alpha, beta1, beta2, eps, 0
        self.state = {}

    def update(self, params, grads, lr_mult=None):
        self.t += 1
        lr_t = self.alpha * math.sqrt(1 - self.beta2 ** self.t) / (1 - self.beta1 ** self.t)
        for k, (p, g) in enumerate(zip(params, grads)):
            if g is None:
                continue
            m, v = self.state.setdefault(k, (np.zeros_like(p), np.zeros_like(p)))
            lr = lr_t * (1.0 if lr_mult is None else lr_mult[k])
            if lr != 0:
                adam_update(p, g, m, v, lr, 1 - self.beta1, 1 - self.beta2, self.eps)
