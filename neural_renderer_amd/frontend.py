"""Fused geometry + lighting front-end of `Renderer.render*` (reference neural_renderer/renderer.py:35-107).

`project_and_light(renderer, vertices, faces, textures)` produces what the reference computes with
fill_back -> vertices_to_faces -> lighting -> look_at / look -> perspective -> vertices_to_faces, i.e. the rasterizer's
inputs `faces [B, F, 3, 3]` and lit `textures [B, F, ts, ts, ts, 3]`, with ONE HIP kernel per direction
(`nr_frontend_forward` / `nr_frontend_backward`, csrc/nr_frontend.hip) instead of ~60 + ~100 small torch launches.
Gradients reach `vertices`, `textures` and a learnable camera position `eye` (example4).

`fusable(...)` decides whether a call fits the kernel's parameter space (CUDA float32 tensors, look_at / look camera,
numeric viewing angle, one light for the whole batch); everything else -- CPU tensors, tensor-valued angles, per-image
light colours -- keeps the module-by-module torch path of renderer.py, which mirrors the reference line by line.
"""
import numpy as np
import torch

from . import _lib, _util


def _vec3(value):
    """Host float32 [3] from a list / tuple / ndarray, or None if `value` is not that (tensor, per-batch array, ...)."""
    if torch.is_tensor(value):
        return None
    try:
        arr = np.asarray(value, dtype=np.float32)
    except (TypeError, ValueError):
        return None
    return arr if arr.shape == (3,) else None


def _number(value):
    return isinstance(value, (int, float, np.integer, np.floating)) and not isinstance(value, bool)


def fusable(renderer, vertices, faces, textures):
    if not (torch.is_tensor(vertices) and vertices.is_cuda and vertices.dtype == torch.float32 and vertices.dim() == 3
            and vertices.shape[2] == 3):
        return False
    if not (torch.is_tensor(faces) and faces.is_cuda and faces.dim() == 3 and faces.shape[2] == 3
            and faces.shape[0] == vertices.shape[0] and not faces.is_floating_point()):
        return False
    if vertices.shape[0] > 65535:
        return False
    if textures is not None:
        if not (torch.is_tensor(textures) and textures.is_cuda and textures.dtype == torch.float32 and textures.dim() == 6
                and tuple(textures.shape[:2]) == tuple(faces.shape[:2]) and textures.shape[5] == 3
                and textures.shape[2] == textures.shape[3] == textures.shape[4]):
            return False
        if not (_number(renderer.light_intensity_ambient) and _number(renderer.light_intensity_directional)):
            return False
        if any(_vec3(v) is None for v in (renderer.light_color_ambient, renderer.light_color_directional,
                                          renderer.light_direction)):
            return False
    if renderer.camera_mode == 'look':
        if _vec3(renderer.camera_direction) is None:
            return False
    elif renderer.camera_mode != 'look_at':
        return False
    if renderer.perspective and not _number(renderer.viewing_angle):
        return False
    eye = renderer.eye
    if torch.is_tensor(eye):
        if eye.dtype != torch.float32 or tuple(eye.shape) not in ((3,), (vertices.shape[0], 3)):
            return False
    else:
        arr = np.asarray(eye, dtype=np.float32)
        if arr.shape not in ((3,), (vertices.shape[0], 3)):
            return False
    return True


_STRUCT_CACHE = {}


def _cached(kind, key, make):
    """Camera / light structs per distinct parameter set: a fixed-shape optimisation loop (example 2, B = 1) is bound by
    host time, and rebuilding two ctypes structs through NumPy costs more than the kernel they parameterise."""
    k = (kind,) + key
    v = _STRUCT_CACHE.get(k)
    if v is None:
        if len(_STRUCT_CACHE) > 128:
            _STRUCT_CACHE.clear()
        v = _STRUCT_CACHE[k] = make()
    return v


def _hashable(v):
    return tuple(np.asarray(v, dtype=np.float64).reshape(-1).tolist())


def _camera_struct(renderer):
    key = (renderer.camera_mode, bool(renderer.perspective),
           float(renderer.viewing_angle) if renderer.perspective else 0.0,
           _hashable(renderer.camera_direction) if renderer.camera_mode == 'look' else ())
    return _cached('camera', key, lambda: _make_camera_struct(renderer))


def _make_camera_struct(renderer):
    cam = _lib.Camera()
    if renderer.camera_mode == 'look_at':
        cam.mode = _lib.NR_CAMERA_LOOK_AT
        target = np.zeros(3, np.float32)  # `at` default, look_at.py:13-14 (Renderer never passes another one)
    else:
        cam.mode = _lib.NR_CAMERA_LOOK
        target = _vec3(renderer.camera_direction)
    cam.perspective = 1 if renderer.perspective else 0
    for k in range(3):
        cam.target[k] = float(target[k])
        cam.up[k] = (0.0, 1.0, 0.0)[k]  # look_at.py:17-18, look.py:16-17
    # perspective.py:10-13 in float32: angle / 180 * 3.1416, tan
    angle = np.float32(renderer.viewing_angle) / np.float32(180.) * np.float32(3.1416) if renderer.perspective \
        else np.float32(0)
    cam.width = float(np.tan(angle, dtype=np.float32))
    return cam


def _light_struct(renderer):
    key = (float(renderer.light_intensity_ambient), float(renderer.light_intensity_directional),
           _hashable(renderer.light_color_ambient), _hashable(renderer.light_color_directional),
           _hashable(renderer.light_direction))
    return _cached('light', key, lambda: _make_light_struct(renderer))


def _make_light_struct(renderer):
    light = _lib.Light()
    light.intensity_ambient = float(renderer.light_intensity_ambient)
    light.intensity_directional = float(renderer.light_intensity_directional)
    for name, src in (('color_ambient', renderer.light_color_ambient),
                      ('color_directional', renderer.light_color_directional), ('direction', renderer.light_direction)):
        v = _vec3(src)
        for k in range(3):
            getattr(light, name)[k] = float(v[k])
    return light


class _FrontEnd(torch.autograd.Function):
    """forward(ctx, vertices, textures | None, eye, faces_idx, camera, light | None, fill_back)
    -> (faces [B,F,3,3], lit textures [B,F,ts,ts,ts,3] | None)."""

    @staticmethod
    def forward(ctx, vertices, textures, eye, faces_idx, camera, light, fill_back):
        lib = _lib.load()
        dev = vertices.device
        v = vertices.detach().contiguous()
        idx = faces_idx.detach().to(torch.int32).contiguous()
        e = eye.detach().contiguous()
        t = textures.detach().contiguous() if textures is not None else None
        B, Nv = v.shape[:2]
        Nf = idx.shape[1]
        F = Nf * 2 if fill_back else Nf
        ts = int(t.shape[2]) if t is not None else 0
        faces_out = torch.empty((B, F, 3, 3), dtype=torch.float32, device=dev)
        textures_out = torch.empty((B, F, ts, ts, ts, 3), dtype=torch.float32, device=dev) if t is not None else None
        with torch.cuda.device(dev):
            _lib.check(lib.nr_frontend_forward(
                v.data_ptr(), idx.data_ptr(), _lib.ptr(t), e.data_ptr(), faces_out.data_ptr(), _lib.ptr(textures_out),
                B, Nv, Nf, ts, 1, int(e.dim() == 2), int(fill_back), camera, light,
                torch.cuda.current_stream(dev).cuda_stream), 'nr_frontend_forward')
        ctx.save_for_backward(v, idx, e, t)
        ctx.params = (camera, light, bool(fill_back), B, Nv, Nf, ts)
        ctx.set_materialize_grads(False)
        return faces_out, textures_out

    @staticmethod
    def backward(ctx, g_faces, g_textures_out):
        lib = _lib.load()
        v, idx, e, t = ctx.saved_tensors
        camera, light, fill_back, B, Nv, Nf, ts = ctx.params
        dev = v.device
        need_v, need_t, need_e = ctx.needs_input_grad[0], ctx.needs_input_grad[1], ctx.needs_input_grad[2]
        F = Nf * 2 if fill_back else Nf
        if g_faces is None:
            g_faces = torch.zeros((B, F, 3, 3), dtype=torch.float32, device=dev)
        g_faces = g_faces.contiguous()
        if g_textures_out is not None:
            g_textures_out = g_textures_out.contiguous()
        need_t = need_t and t is not None and g_textures_out is not None
        need_v = need_v or need_e  # the camera sums come out of the vertex pass
        grad_v = torch.empty((B, Nv, 3), dtype=torch.float32, device=dev) if need_v else None
        grad_t = torch.empty_like(t) if need_t else None
        grad_e = torch.empty_like(e) if need_e else None
        if not (need_v or need_t):
            return None, None, None, None, None, None, None
        ws_bytes = lib.nr_frontend_workspace_bytes(B) if need_e else 0
        ws = torch.empty((max(ws_bytes, 1),), dtype=torch.uint8, device=dev)
        with torch.cuda.device(dev):
            _lib.check(lib.nr_frontend_backward(
                v.data_ptr(), idx.data_ptr(), _lib.ptr(t), e.data_ptr(), g_faces.data_ptr(),
                _lib.ptr(g_textures_out) if t is not None else None, _lib.ptr(grad_v), _lib.ptr(grad_t), _lib.ptr(grad_e),
                B, Nv, Nf, ts, 1, int(e.dim() == 2), int(fill_back), camera, light, ws.data_ptr(), ws_bytes,
                torch.cuda.current_stream(dev).cuda_stream), 'nr_frontend_backward')
        return (grad_v if ctx.needs_input_grad[0] else None), grad_t, grad_e, None, None, None, None


class _FrontEndLight(torch.autograd.Function):
    """forward(ctx, vertices, eye, faces_idx, camera, light, fill_back) -> (faces [B,F,3,3], light colours [B,F,3]):
    the front-end for the rasterizer's `face_light` mode (include/nr_hip.h: nr_frontend_forward_light) -- no textures pass
    through; their gradient comes straight out of the rasterizer."""

    @staticmethod
    def forward(ctx, vertices, eye, faces_idx, camera, light, fill_back):
        lib = _lib.load()
        dev = vertices.device
        v = vertices.detach().contiguous()
        idx = faces_idx.detach().to(torch.int32).contiguous()
        e = eye.detach().contiguous()
        B, Nv = v.shape[:2]
        Nf = idx.shape[1]
        F = Nf * 2 if fill_back else Nf
        faces_out = torch.empty((B, F, 3, 3), dtype=torch.float32, device=dev)
        light_out = torch.empty((B, F, 3), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            _lib.check(lib.nr_frontend_forward_light(
                v.data_ptr(), idx.data_ptr(), e.data_ptr(), faces_out.data_ptr(), light_out.data_ptr(), B, Nv, Nf, 1,
                int(e.dim() == 2), int(fill_back), camera, light, torch.cuda.current_stream(dev).cuda_stream),
                'nr_frontend_forward_light')
        ctx.save_for_backward(v, idx, e)
        ctx.params = (camera, light, bool(fill_back), B, Nv, Nf)
        ctx.set_materialize_grads(False)
        return faces_out, light_out

    @staticmethod
    def backward(ctx, g_faces, g_light):
        lib = _lib.load()
        v, idx, e = ctx.saved_tensors
        camera, light, fill_back, B, Nv, Nf = ctx.params
        dev = v.device
        need_v, need_e = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        if not (need_v or need_e):
            return None, None, None, None, None, None
        F = Nf * 2 if fill_back else Nf
        if g_faces is None:
            g_faces = torch.zeros((B, F, 3, 3), dtype=torch.float32, device=dev)
        g_faces = g_faces.contiguous()
        if g_light is not None:
            g_light = g_light.contiguous()
        grad_v = torch.empty((B, Nv, 3), dtype=torch.float32, device=dev)  # the camera sums come out of the vertex pass
        grad_e = torch.empty_like(e) if need_e else None
        ws_bytes = lib.nr_frontend_workspace_bytes(B) if need_e else 0
        ws = torch.empty((max(ws_bytes, 1),), dtype=torch.uint8, device=dev)
        with torch.cuda.device(dev):
            _lib.check(lib.nr_frontend_backward_light(
                v.data_ptr(), idx.data_ptr(), e.data_ptr(), g_faces.data_ptr(), _lib.ptr(g_light), grad_v.data_ptr(),
                _lib.ptr(grad_e), B, Nv, Nf, 1, int(e.dim() == 2), int(fill_back), camera, light, ws.data_ptr(), ws_bytes,
                torch.cuda.current_stream(dev).cuda_stream), 'nr_frontend_backward_light')
        return (grad_v if need_v else None), grad_e, None, None, None, None


_EYE_CACHE = {}


def _eye_tensor(eye, device):
    if torch.is_tensor(eye):
        return eye.to(device=device)
    arr = np.ascontiguousarray(eye, dtype=np.float32)
    key = (str(device), arr.shape, arr.tobytes())
    t = _EYE_CACHE.get(key)
    if t is None:
        if len(_EYE_CACHE) > 64:
            _EYE_CACHE.clear()
        t = _EYE_CACHE[key] = torch.as_tensor(arr, device=device)
    return t


def project_and_light(renderer, vertices, faces, textures=None):
    """-> (faces [B,F,3,3], lit textures | None) for the rasterizer; call only when fusable(...)."""
    _util.check_face_indices(faces, vertices.shape[1], vertices.device)
    camera = _camera_struct(renderer)
    light = _light_struct(renderer) if textures is not None else None
    eye = _eye_tensor(renderer.eye, vertices.device)
    return _FrontEnd.apply(vertices, textures, eye, faces, camera, light, bool(renderer.fill_back))


def project_and_light_colors(renderer, vertices, faces):
    """-> (faces [B,F,3,3], light colours [B,F,3]) for the rasterizer's face_light mode; call only when fusable(...)."""
    _util.check_face_indices(faces, vertices.shape[1], vertices.device)
    eye = _eye_tensor(renderer.eye, vertices.device)
    return _FrontEndLight.apply(vertices, eye, faces, _camera_struct(renderer), _light_struct(renderer),
                                bool(renderer.fill_back))
