"""Rasterizer operator: the reference's `neural_renderer/rasterize.py` API on PyTorch-ROCm tensors,
backed by the hand-written HIP kernels of libnr_hip.so (include/nr_hip.h) through ctypes.

Mirrors (reference file:line):
  Rasterize(image_size, near, far, eps, background_color, return_rgb, return_alpha, return_depth)
      rasterize.py:19-64 (constructor), :467-513 (forward_gpu), :849-889 (backward_gpu)
  rasterize_rgbad / rasterize / rasterize_silhouettes / rasterize_depth      rasterize.py:900-1060
  use_unsafe_rasterizer                                                        rasterize.py:1063-1065
  module defaults                                                              rasterize.py:7-16

There is no CPU path (the reference has none either: rasterize.py:893-897) and no eager fallback.
"""
import collections
import os
import threading
import weakref

import numpy as np
import torch

from . import _lib

DEFAULT_IMAGE_SIZE = 256
DEFAULT_ANTI_ALIASING = True
DEFAULT_NEAR = 0.1
DEFAULT_FAR = 100
DEFAULT_EPS = 1e-4
DEFAULT_BACKGROUND_COLOR = (0, 0, 0)
USE_UNSAFE_IMPLEMENTATION = False

if 'NEURAL_RENDERER_UNSAFE' in os.environ and int(os.environ['NEURAL_RENDERER_UNSAFE']):
    USE_UNSAFE_IMPLEMENTATION = True

# the reference samples textures with the face depth of batch element 0
# (rasterize.py:389).  Default = reference-literal; NR_FIX_TEXTURE_BATCH_Z=1 (or the `fix_batch_z`
# attribute of a Rasterize instance) uses the pixel's own batch element.
FIX_TEXTURE_BATCH_Z = bool(int(os.environ.get('NR_FIX_TEXTURE_BATCH_Z', '0')))

# K6 (backward_pixel_map) numerics.  Default: float terms through fused multiply-adds and the hardware reciprocal, ~1 ulp per
# term; the tests bound the deviation from the reference's terms summed exactly by the north star's 1e-4 (plus twice the reference's
# own float-summation noise on entries that cancel down to the metric's floor: include/nr_hip.h).  NR_EXACT_GRADIENT=1
# (read once, here) or the `exact_gradient` attribute of a Rasterize instance: every term with the reference's own arithmetic,
# sums in double (bound 2e-6).  Measured levels and costs of both: profiles/*_parity_summary.md.
# Reproducibility: both modes return the same grad_faces bit for bit from call to call, and for a batch and its shards (one band
# kernel, k_bpm_row, whose per-record sums do not depend on the launch; profiles/r06_same_terms.txt).
EXACT_GRADIENT = bool(int(os.environ.get('NR_EXACT_GRADIENT', '0')))
# (measuring aid, read once: NR_SERIAL_BACKWARD=1 makes the fused backward launch K6's line setup, its band kernel and the gather one
# after the other instead of the first and the last in one grid -- include/nr_hip.h NR_FLAG_SERIAL_BACKWARD; same values)
_BACKWARD_ORDER_FLAG = _lib.NR_FLAG_SERIAL_BACKWARD if int(os.environ.get('NR_SERIAL_BACKWARD', '0')) else 0
# (measuring aid, read once: NR_K6_LEGACY=1 keeps K6's default mode on the piece-per-lane band kernel -- NR_FLAG_K6_LEGACY)
_BACKWARD_ORDER_FLAG |= _lib.NR_FLAG_K6_LEGACY if int(os.environ.get('NR_K6_LEGACY', '0')) else 0


_RAW_STREAM = getattr(torch._C, '_cuda_getCurrentRawStream', None)


def _stream_ptr(device):
    """hipStream_t of torch's current stream on `device` (the raw query: no Stream object per call)."""
    if _RAW_STREAM is not None:
        return _RAW_STREAM(device.index)
    return torch.cuda.current_stream(device).cuda_stream


class _NoGuard(object):
    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


_NO_GUARD = _NoGuard()


def _on_device(device):
    """Context that makes `device` the current HIP device for the launches -- nothing at all when it already is (one process
    per GPU: always), torch.cuda.device otherwise."""
    return _NO_GUARD if torch.cuda.current_device() == device.index else torch.cuda.device(device)


_BG_CACHE = {}


def _background_tensor(bg, device):
    """Device copy of a host background colour, cached (a pageable H2D copy per call would serialise the stream)."""
    # (the fast key only for tuples of plain Python numbers: a 0-d tensor or array element hashes by identity and may change in
    # place; floats are keyed by their repr so that -0.0 and 0.0 stay apart)
    fast = type(bg) is tuple and all(type(x) in (int, float) for x in bg)
    if fast:
        key = (device.index, tuple(repr(x) for x in bg))
        t = _BG_CACHE.get(key)
        if t is not None:
            return t
    arr = np.asarray(bg, dtype=np.float32)
    key2 = (device.index, arr.shape, arr.tobytes())
    t = _BG_CACHE.get(key2)
    if t is None:
        if len(_BG_CACHE) > 64:
            _BG_CACHE.clear()
        t = _BG_CACHE[key2] = torch.as_tensor(arr, device=device)
    if fast:
        _BG_CACHE[key] = t
    return t


# Graph replay (not in the reference): for fixed shapes the forward and the backward of the operator are each captured once
# in a HIP graph and replayed from then on -- at small batches the ~12 launches of a step take longer to ISSUE than to run
# (BASELINE config 2, 16 views: 0.27 ms eager, 0.18 ms replayed).  Off by default; `use_graph_replay(True)`, the
# NR_GRAPH_REPLAY=1 environment variable (read once, here) or the `graph_replay` attribute of a Rasterize / Renderer
# instance switch it on.  Semantics stay those of the eager operator: inputs are copied into the graphs' fixed buffers and
# results are returned as copies, so callers may keep them; what replay cannot offer is two forwards of the same shapes in
# flight before the first one's backward (the residual maps live in the fixed buffers) -- that raises.  Measured (config 2):
# 0.25 ms eager -> 0.21 ms with replay; a step captured as a whole by the caller (neural_renderer_amd.graph.capture: no
# copies, loss and optimizer inside) 0.17 ms -- prefer that where the whole step is fixed.  (Round 3's crash of a whole-step
# capture after this mode had run is gone: neural_renderer_amd.graph.capture.)  `clear_graph_replay_cache()` drops the graphs.
GRAPH_REPLAY = bool(int(os.environ.get('NR_GRAPH_REPLAY', '0')))


def use_graph_replay(flag):
    global GRAPH_REPLAY
    GRAPH_REPLAY = bool(flag)


# The forward's scratch (z-buffer + queues), kept per (device, stream, sizes) and reused with a falling epoch number
# (include/nr_hip.h: NR_FLAG_ZBUF_EPOCH) so that the library neither fills the z-buffer before a call nor cleans it afterwards.
# Least-recently-used entries go first; the cache holds at most _ZBUF_CACHE_BYTES (a workspace larger than half of that is
# not kept at all: such a call takes the per-call fill); `clear_workspace_cache()` releases everything.
_ZBUF_CACHE = collections.OrderedDict()
_ZBUF_CACHE_BYTES = int(os.environ.get('NR_WORKSPACE_CACHE_MB', '1024')) << 20
_ZBUF_LOCK = threading.Lock()


def clear_workspace_cache():
    """Release the kept forward workspaces (8 bytes per raster pixel each; see _forward_workspace)."""
    with _ZBUF_LOCK:
        _ZBUF_CACHE.clear()


def _forward_workspace(lib, dev, stream, B, F, S):
    """(workspace tensor, bytes, flags) for one forward call.  A kept workspace is refilled with 0xff every 255 calls.  While
    a HIP graph is being captured the epoch would be frozen into the graph, so capture takes a throw-away workspace and the
    filling path."""
    ws_bytes = lib.nr_forward_workspace_bytes(B, F, S)
    if ws_bytes == 0:
        raise ValueError('unsupported sizes B=%d F=%d S=%d' % (B, F, S))
    if F >= (1 << 24) or 2 * ws_bytes > _ZBUF_CACHE_BYTES or torch.cuda.is_current_stream_capturing():
        return torch.empty((ws_bytes,), dtype=torch.uint8, device=dev), ws_bytes, 0
    # (the calling THREAD is part of the key: the epochs of a workspace must reach the stream in the order they were handed out
    # -- a call launched later with the larger epoch of an earlier hand-out would lose every atomic minimum against the words of
    # the call in front of it -- and within one host thread hand-out order is launch order; two threads rendering the same shapes
    # on one stream each keep their own workspace)
    key = (dev.index, int(stream), B, F, S, threading.get_ident())
    with _ZBUF_LOCK:
        ent = _ZBUF_CACHE.get(key)
        if ent is None:
            total = sum(e[0].numel() for e in _ZBUF_CACHE.values()) + ws_bytes
            while _ZBUF_CACHE and (total > _ZBUF_CACHE_BYTES or len(_ZBUF_CACHE) >= 16):
                total -= _ZBUF_CACHE.popitem(last=False)[1][0].numel()
            ent = _ZBUF_CACHE[key] = [torch.empty((ws_bytes,), dtype=torch.uint8, device=dev), -1]
        else:
            _ZBUF_CACHE.move_to_end(key)
        if ent[1] < 0:
            ent[0].fill_(255)
            ent[1] = 254
        epoch = ent[1]
        ent[1] -= 1
    return ent[0], ws_bytes, _lib.NR_FLAG_ZBUF_EPOCH | (epoch << 8)


class _Config(object):
    """What a Rasterize instance is configured with, frozen for one call (the autograd node keeps it)."""
    __slots__ = ('image_size', 'near', 'far', 'eps', 'background_color', 'return_rgb', 'return_alpha', 'return_depth',
                 'fix_batch_z', 'exact_gradient', 'faces_z_ref', 'owner')

    def __init__(self, fn):
        self.image_size = int(fn.image_size)
        self.near, self.far, self.eps = float(fn.near), float(fn.far), float(fn.eps)
        self.background_color = fn.background_color if fn.background_color is not None else DEFAULT_BACKGROUND_COLOR
        self.return_rgb, self.return_alpha, self.return_depth = bool(fn.return_rgb), bool(fn.return_alpha), bool(fn.return_depth)
        self.fix_batch_z, self.exact_gradient = bool(fn.fix_batch_z), bool(fn.exact_gradient)
        self.faces_z_ref = fn.faces_z_ref
        self.owner = weakref.ref(fn)


class _Residuals(object):
    """Everything one forward leaves behind for its backward (the reference keeps the same on `self`, rasterize.py:39-58)."""
    __slots__ = ('B', 'F', 'S', 'ts', 'Nf', 'flags', 'faces', 'textures', 'light', 'z_ref', 'face_index_map', 'weight_map',
                 'depth_map', 'rgb_map', 'alpha_map', 'visible')


def _check_inputs(cfg, faces, textures, light):
    """Type / shape checks of rasterize.py:66-90 (+ the face_light extension).  Returns (B, F, Nf, ts)."""
    if not faces.is_cuda:
        raise NotImplementedError('neural_renderer_amd has no CPU rasterizer (neither has the reference: '
                                  'rasterize.py:893-897)')
    if faces.dtype != torch.float32 or faces.dim() != 4 or tuple(faces.shape[2:]) != (3, 3):
        raise ValueError('faces must be float32 [batch size, num of faces, 3, 3], got %s %s'
                         % (faces.dtype, tuple(faces.shape)))
    B, F = int(faces.shape[0]), int(faces.shape[1])
    Nf, ts = F, 0
    if cfg.return_rgb:
        if textures is None:
            raise ValueError('textures are required when return_rgb is set')
        if light is not None:
            if light.dtype != torch.float32 or tuple(light.shape) != (B, F, 3):
                raise ValueError('face_light must be float32 [batch size, num of faces, 3], got %s %s'
                                 % (light.dtype, tuple(light.shape)))
            if textures.dim() == 6 and textures.shape[1] * 2 == F:
                Nf = F // 2  # fill_back: face Nf + f is the reversed copy of face f
        sh = textures.shape
        if (textures.dtype != torch.float32 or textures.dim() != 6 or sh[0] != B or sh[1] != Nf or sh[2] < 2 or
                sh[2] != sh[3] or sh[3] != sh[4] or sh[5] != 3):
            raise ValueError('textures must be float32 [batch size, num of faces, ts, ts, ts, 3] with ts >= 2, '
                             'got %s %s' % (textures.dtype, tuple(sh)))  # rasterize.py:78-90
        ts = int(sh[2])
    return B, F, Nf, ts


def _forward_impl(cfg, faces, textures, light):
    """forward_gpu (rasterize.py:467-513): visibility + shading behind one C-ABI call.  Returns the residuals."""
    lib = _lib.load()
    B, F, Nf, ts = _check_inputs(cfg, faces, textures, light)
    return_rgb, return_alpha, return_depth = cfg.return_rgb, cfg.return_alpha, cfg.return_depth
    dev = faces.device
    S = cfg.image_size
    r = _Residuals()
    r.B, r.F, r.S, r.ts, r.Nf = B, F, S, ts, Nf
    r.faces = faces.detach().contiguous()  # rasterize.py:470
    r.textures = textures.detach().contiguous() if return_rgb else None  # :473
    r.light = light.detach().contiguous() if (return_rgb and light is not None) else None
    with _on_device(dev):
        stream = _stream_ptr(dev)
        need_wd = return_rgb or return_depth
        i32, f32 = torch.int32, torch.float32
        r.face_index_map = torch.empty((B, S, S), dtype=i32, device=dev)
        r.weight_map = torch.empty((B, S, S, 3), dtype=f32, device=dev) if need_wd else None
        r.depth_map = torch.empty((B, S, S), dtype=f32, device=dev) if need_wd else None
        workspace, ws_bytes, ws_flags = _forward_workspace(lib, dev, stream, B, F, S)
        r.rgb_map = r.alpha_map = background = None
        bg_per_batch = 0
        if return_rgb:
            r.rgb_map = torch.empty((B, S, S, 3), dtype=f32, device=dev)
            bg = cfg.background_color
            if torch.is_tensor(bg):
                background = bg.detach().to(device=dev, dtype=f32).contiguous()
            else:
                background = _background_tensor(bg, dev)
            if tuple(background.shape) == (B, 3):
                bg_per_batch = 1  # rasterize.py:464-465
            elif tuple(background.shape) != (3,):
                raise ValueError('background_color must have shape (3,) or (batch size, 3)')
        if return_alpha:
            r.alpha_map = torch.empty((B, S, S), dtype=f32, device=dev)
        flags = (_lib.NR_FLAG_FIX_TEXTURE_BATCH_Z if cfg.fix_batch_z else 0) | _BACKWARD_ORDER_FLAG
        if cfg.exact_gradient:
            flags |= _lib.NR_FLAG_EXACT_GRADIENT
        r.flags = flags
        # batch element 0 of the GLOBAL batch when this call holds a shard of it (include/nr_hip.h)
        z_ref = cfg.faces_z_ref
        if z_ref is not None:
            z_ref = z_ref.detach().to(device=dev, dtype=f32).contiguous()
            if tuple(z_ref.shape) != (F, 3, 3):
                raise ValueError('faces_z_ref must have shape (num of faces, 3, 3), got %s' % (tuple(z_ref.shape),))
        r.z_ref = z_ref
        # per-face "owns a pixel" flags: a residual the backward starts from (K6's lists; the depth-only gather skips the
        # faces without a pixel)
        r.visible = torch.empty((B, F), dtype=torch.uint8, device=dev)
        ptr = _lib.ptr
        lit = _lib.FaceLight(r.light.data_ptr(), Nf, None, None) if r.light is not None else None
        # visibility + shading behind one call (rasterize.py:499-502).  weight_map is a residual only (the backward reads it at
        # covered pixels): the zeros of uncovered pixels are not stored (NR_FLAG_SPARSE_WEIGHT_MAP; `Rasterize.weight_map`
        # fills them in when somebody reads the attribute)
        _lib.check(lib.nr_forward_rasterize_lit(
            lit, r.faces.data_ptr(), ptr(z_ref), ptr(r.textures), r.face_index_map.data_ptr(), ptr(r.weight_map),
            ptr(r.depth_map), ptr(r.rgb_map), ptr(r.alpha_map), r.visible.data_ptr(), ptr(background), bg_per_batch,
            B, F, S, ts, cfg.near, cfg.far, cfg.eps, flags | ws_flags | _lib.NR_FLAG_SPARSE_WEIGHT_MAP,
            workspace.data_ptr(), ws_bytes, stream), 'nr_forward_rasterize')
    return r


_BWD_WS_BYTES = {}


def _backward_impl(cfg, r, g_rgb, g_alpha, g_depth, want_textures, want_light):
    """backward_gpu (rasterize.py:849-889): K6 -> K7 -> K8 behind one C-ABI call.  `None` gradients are zeros (:858-878); a
    zero gradient adds exactly 0 to every `diff_grad`, so the corresponding term is skipped instead of being multiplied out.
    Returns (grad_faces, grad_textures | None, grad_light | None)."""
    lib = _lib.load()
    B, F, S, ts = r.B, r.F, r.S, r.ts
    use_rgb = cfg.return_rgb and g_rgb is not None
    use_alpha = cfg.return_alpha and g_alpha is not None
    use_depth = cfg.return_depth and g_depth is not None
    if not (use_rgb or use_alpha or use_depth):
        return None, None, None
    dev = r.faces.device
    grad_textures = grad_light = None
    with _on_device(dev):
        stream = _stream_ptr(dev)
        g_rgb = g_rgb.contiguous() if use_rgb else None
        g_alpha = g_alpha.contiguous() if use_alpha else None
        g_depth = g_depth.contiguous() if use_depth else None
        grad_faces = torch.empty_like(r.faces)  # stored by the library (zeros when neither rgb nor alpha)
        lit = None
        f32 = torch.float32
        if use_rgb and r.light is not None:
            if want_textures or want_light:
                # one gather produces both (the colours' gradient is a by-product of the texel sums)
                grad_textures = torch.empty((B, r.Nf, ts, ts, ts, 3), dtype=f32, device=dev)
                if want_light:
                    grad_light = torch.empty((B, F, 3), dtype=f32, device=dev)
            lit = _lib.FaceLight(r.light.data_ptr(), r.Nf, r.textures.data_ptr(), _lib.ptr(grad_light))
        elif use_rgb and want_textures:
            grad_textures = torch.empty((B, F, ts, ts, ts, 3), dtype=f32, device=dev)
        key = (B, F, S, use_rgb, use_alpha)
        ws_bytes = _BWD_WS_BYTES.get(key)
        if ws_bytes is None:
            if len(_BWD_WS_BYTES) > 256:
                _BWD_WS_BYTES.clear()
            ws_bytes = _BWD_WS_BYTES[key] = lib.nr_backward_workspace_bytes(B, F, S, int(use_rgb), int(use_alpha))
        workspace = torch.empty((max(ws_bytes, 1),), dtype=torch.uint8, device=dev)
        ptr = _lib.ptr
        # K6 -> K7 -> K8 (rasterize.py:881-883) behind one call
        _lib.check(lib.nr_backward_rasterize_lit(
            lit, r.faces.data_ptr(), ptr(r.z_ref), r.face_index_map.data_ptr(), ptr(r.weight_map), ptr(r.depth_map),
            ptr(r.rgb_map) if use_rgb else None, ptr(r.alpha_map) if use_alpha else None, ptr(g_rgb), ptr(g_alpha),
            ptr(g_depth), grad_faces.data_ptr(), ptr(grad_textures), B, F, S, ts, cfg.eps, r.flags, ptr(r.visible),
            workspace.data_ptr(), ws_bytes, stream), 'nr_backward_rasterize')
    owner = cfg.owner()
    if owner is not None:  # rasterize.py:41-51: the gradient buffers stay readable on the instance
        owner.grad_rgb_map, owner.grad_alpha_map, owner.grad_depth_map = g_rgb, g_alpha, g_depth
        # (aliases, not the returned tensors themselves: a second reference to a returned gradient makes autograd's
        # AccumulateGrad clone it instead of adopting it -- two device copies, 41 MB per step at the headline size)
        owner.grad_faces = grad_faces.detach()
        owner.grad_textures = grad_textures.detach() if grad_textures is not None else None
    return grad_faces, grad_textures, grad_light


class _RasterizeFunction(torch.autograd.Function):
    """forward(ctx, faces, textures, cfg, light) -> (rgb_map [B,S,S,3] | None, alpha_map [B,S,S] | None,
    depth_map [B,S,S] | None, face_index_map); backward(ctx, g_rgb, g_alpha, g_depth, _) -> (grad_faces, grad_textures, None,
    grad_light).  `light` [B,F,3] (or None): per-face light colours, textures are then the original cubes [B,Nf,...] with
    F = Nf or 2 Nf (include/nr_hip.h: nr_face_light)."""

    @staticmethod
    def forward(ctx, faces, textures, cfg, light=None):
        if not cfg.return_rgb:
            textures = light = None
        r = _forward_impl(cfg, faces, textures, light)
        owner = cfg.owner()
        if owner is not None:
            owner._keep(r)
        ctx.cfg = cfg
        # (the node must not hold its own OUTPUTS except through save_for_backward: output -> grad_fn -> node -> output is a
        # cycle through C++ that nothing collects -- every step's maps would stay allocated)
        ctx.meta = (r.B, r.F, r.S, r.ts, r.Nf, r.flags)
        ctx.z_ref, ctx.visible = r.z_ref, r.visible
        ctx.set_materialize_grads(False)  # an unused output arrives as `None` in backward and its terms are skipped
        # residuals (the reference keeps them on `self`, rasterize.py:39-58); outputs and inputs among them go through
        # save_for_backward so that in-place edits by the caller are detected
        ctx.save_for_backward(r.faces, r.face_index_map, r.weight_map, r.depth_map, r.rgb_map, r.alpha_map, r.textures if
                              r.light is not None else None, r.light)
        ctx.mark_non_differentiable(r.face_index_map)
        return (r.rgb_map if cfg.return_rgb else None, r.alpha_map if cfg.return_alpha else None,
                r.depth_map if cfg.return_depth else None, r.face_index_map)

    @staticmethod
    def backward(ctx, g_rgb, g_alpha, g_depth, _g_fi):
        r = _Residuals()
        r.B, r.F, r.S, r.ts, r.Nf, r.flags = ctx.meta
        r.z_ref, r.visible = ctx.z_ref, ctx.visible
        # (unpacking checks the version counters of the saved tensors: an in-place edit since the forward raises)
        r.faces, r.face_index_map, r.weight_map, r.depth_map, r.rgb_map, r.alpha_map, r.textures, r.light = ctx.saved_tensors
        need = ctx.needs_input_grad
        gf, gt, gl = _backward_impl(ctx.cfg, r, g_rgb, g_alpha, g_depth, need[1], need[3])
        return gf, (gt if need[1] else None), None, gl


def _capture(fn, dev):
    """Warm `fn` up on a side stream, then capture it (see neural_renderer_amd.graph.capture)."""
    from .graph import capture
    return capture(fn, dev, warmup=2)


class _GraphEntry(object):
    """Fixed buffers + captured graphs of one (device, sizes, configuration) of the operator."""

    def __init__(self, lib, dev, cfg, B, F, S, ts, bg, bg_per_batch, has_z_ref):
        self.lib, self.dev, self.cfg = lib, dev, cfg
        self.dims = (B, F, S, ts)
        f32 = dict(dtype=torch.float32, device=dev)
        rgb, alpha, depth = cfg.return_rgb, cfg.return_alpha, cfg.return_depth
        need_wd = rgb or depth
        self.faces = torch.zeros((B, F, 3, 3), **f32)
        self.textures = torch.zeros((B, F, ts, ts, ts, 3), **f32) if rgb else None
        self.z_ref = torch.zeros((F, 3, 3), **f32) if has_z_ref else None
        self.background = bg.clone() if bg is not None else None
        self.bg_per_batch = bg_per_batch
        self.face_index_map = torch.empty((B, S, S), dtype=torch.int32, device=dev)
        self.weight_map = torch.empty((B, S, S, 3), **f32) if need_wd else None
        self.depth_map = torch.empty((B, S, S), **f32) if need_wd else None
        self.rgb_map = torch.empty((B, S, S, 3), **f32) if rgb else None
        self.alpha_map = torch.empty((B, S, S), **f32) if alpha else None
        self.visible = torch.empty((B, F), dtype=torch.uint8, device=dev)
        ws_bytes = lib.nr_forward_workspace_bytes(B, F, S)
        if ws_bytes == 0:
            raise ValueError('unsupported sizes B=%d F=%d S=%d' % (B, F, S))
        self.fwd_ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=dev)
        self.flags = (_lib.NR_FLAG_FIX_TEXTURE_BATCH_Z if cfg.fix_batch_z else 0) | _BACKWARD_ORDER_FLAG | \
                     (_lib.NR_FLAG_EXACT_GRADIENT if cfg.exact_gradient else 0)
        self.generation = 0
        self.pending = None  # generation of the forward whose residuals the buffers hold and whose backward may still come
        self.bwd = {}        # (use_rgb, use_alpha, use_depth, want grad_textures) -> (graph, buffers)
        self.eager_bwd = {}  # the same for combinations the forward did not foresee: (launcher, buffers)
        with torch.cuda.device(dev):
            self.fwd = _capture(self._forward, dev)

    def _forward(self):
        B, F, S, ts = self.dims
        cfg = self.cfg
        _lib.check(self.lib.nr_forward_rasterize(
            self.faces.data_ptr(), _lib.ptr(self.z_ref), _lib.ptr(self.textures), self.face_index_map.data_ptr(),
            _lib.ptr(self.weight_map), _lib.ptr(self.depth_map), _lib.ptr(self.rgb_map), _lib.ptr(self.alpha_map),
            self.visible.data_ptr(), _lib.ptr(self.background), self.bg_per_batch, B, F, S, ts, cfg.near,
            cfg.far, cfg.eps, self.flags, self.fwd_ws.data_ptr(), self.fwd_ws.numel(),
            _stream_ptr(self.dev)), 'nr_forward_rasterize')

    def _backward_buffers(self, use_rgb, use_alpha, use_depth, want_gt):
        B, F, S, ts = self.dims
        f32 = dict(dtype=torch.float32, device=self.dev)
        buf = {'g_rgb': torch.zeros((B, S, S, 3), **f32) if use_rgb else None,
               'g_alpha': torch.zeros((B, S, S), **f32) if use_alpha else None,
               'g_depth': torch.zeros((B, S, S), **f32) if use_depth else None,
               'grad_faces': torch.empty((B, F, 3, 3), **f32),
               'grad_textures': torch.empty((B, F, ts, ts, ts, 3), **f32) if want_gt else None}
        ws_bytes = self.lib.nr_backward_workspace_bytes(B, F, S, int(use_rgb), int(use_alpha))
        buf['ws'] = torch.empty((max(ws_bytes, 1),), dtype=torch.uint8, device=self.dev)

        def run():
            _lib.check(self.lib.nr_backward_rasterize(
                self.faces.data_ptr(), _lib.ptr(self.z_ref), self.face_index_map.data_ptr(), _lib.ptr(self.weight_map),
                _lib.ptr(self.depth_map), _lib.ptr(self.rgb_map) if use_rgb else None,
                _lib.ptr(self.alpha_map) if use_alpha else None, _lib.ptr(buf['g_rgb']), _lib.ptr(buf['g_alpha']),
                _lib.ptr(buf['g_depth']), buf['grad_faces'].data_ptr(), _lib.ptr(buf['grad_textures']), B, F, S, ts,
                self.cfg.eps, self.flags, self.visible.data_ptr(), buf['ws'].data_ptr(), ws_bytes,
                _stream_ptr(self.dev)), 'nr_backward_rasterize')
        return run, buf

    def prepare_backward(self, key):
        """Capture the backward for one combination of present gradients -- on the CALLER's thread, from the forward: autograd
        runs `backward` on its own device thread, and a graph captured there left the process in a state in which a later
        capture on the main thread crashed (ROCm 7.2 / torch 2.10)."""
        if key not in self.bwd:
            run, buf = self._backward_buffers(*key)
            with torch.cuda.device(self.dev):
                self.bwd[key] = (_capture(run, self.dev), buf)

    def backward_runner(self, key):
        """(callable, buffers): the captured graph when the forward prepared this combination, else plain launches."""
        ent = self.bwd.get(key)
        if ent is None:
            ent = self.eager_bwd.get(key)
            if ent is None:
                ent = self.eager_bwd[key] = self._backward_buffers(*key)
        return ent


_GRAPH_CACHE = {}


def clear_graph_replay_cache():
    """Drop the operator's captured graphs and their fixed buffers (a later call in replay mode captures again)."""
    _GRAPH_CACHE.clear()


class _GraphedRasterizeFunction(torch.autograd.Function):
    """The operator replayed from captured graphs (see GRAPH_REPLAY): same results as _RasterizeFunction."""

    @staticmethod
    def forward(ctx, faces, textures, entry):
        cfg = entry.cfg
        entry.faces.copy_(faces.detach())
        if entry.textures is not None:
            entry.textures.copy_(textures.detach())
        entry.fwd()
        entry.generation += 1
        entry.pending = entry.generation
        ctx.entry, ctx.generation = entry, entry.generation
        ctx.set_materialize_grads(False)
        fi = entry.face_index_map.clone()
        ctx.mark_non_differentiable(fi)
        return (entry.rgb_map.clone() if cfg.return_rgb else None, entry.alpha_map.clone() if cfg.return_alpha else None,
                entry.depth_map.clone() if cfg.return_depth else None, fi)

    @staticmethod
    def backward(ctx, g_rgb, g_alpha, g_depth, _g_fi):
        entry, cfg = ctx.entry, ctx.entry.cfg
        if entry.generation != ctx.generation:
            raise RuntimeError('graph replay: a later forward of the same shapes has replaced the residual maps of this call; '
                               'run backward before the next forward of these shapes, or switch graph replay off')
        use_rgb = cfg.return_rgb and g_rgb is not None
        use_alpha = cfg.return_alpha and g_alpha is not None
        use_depth = cfg.return_depth and g_depth is not None
        if not (use_rgb or use_alpha or use_depth):
            return None, None, None
        want_gt = bool(use_rgb and ctx.needs_input_grad[1])
        graph, buf = entry.backward_runner((use_rgb, use_alpha, use_depth, want_gt))
        for name, g in (('g_rgb', g_rgb if use_rgb else None), ('g_alpha', g_alpha if use_alpha else None),
                        ('g_depth', g_depth if use_depth else None)):
            if g is not None:
                buf[name].copy_(g)
        graph()
        return buf['grad_faces'].clone(), (buf['grad_textures'].clone() if want_gt else None), None


def _graph_entry(faces, textures, cfg):
    """The cached graphs for this call's device / sizes / configuration, or None when the call is not eligible (CPU tensors
    raise in the eager operator; tensor-valued backgrounds or z references that change between calls are handled by value)."""
    if not faces.is_cuda or faces.dtype != torch.float32 or faces.dim() != 4 or torch.cuda.is_current_stream_capturing():
        return None
    dev = faces.device
    B, F = int(faces.shape[0]), int(faces.shape[1])
    S = cfg.image_size
    ts = 0
    bg, bg_per_batch, bg_key = None, 0, None
    if cfg.return_rgb:
        if textures is None or textures.dim() != 6 or textures.dtype != torch.float32 or tuple(textures.shape[:2]) != (B, F):
            return None
        ts = int(textures.shape[2])
        b = cfg.background_color
        if torch.is_tensor(b):
            return None  # a device-side background may change between calls: eager path
        arr = np.asarray(b, dtype=np.float32)
        if arr.shape == (B, 3):
            bg_per_batch = 1
        elif arr.shape != (3,):
            return None
        bg, bg_key = _background_tensor(b, dev), arr.tobytes()
    if cfg.faces_z_ref is not None:
        return None  # sharded batches hand a reference view over per call: eager path
    key = (dev.index, B, F, S, ts, cfg.return_rgb, cfg.return_alpha, cfg.return_depth, cfg.near, cfg.far, cfg.eps, bg_key,
           bg_per_batch, cfg.fix_batch_z, cfg.exact_gradient)
    entry = _GRAPH_CACHE.get(key)
    if entry is None:
        if len(_GRAPH_CACHE) >= 8:
            _GRAPH_CACHE.pop(next(iter(_GRAPH_CACHE)))
        entry = _GRAPH_CACHE[key] = _GraphEntry(_lib.load(), dev, cfg, B, F, S, ts, bg, bg_per_batch, False)
    return entry


class _ImageEpilogue(torch.autograd.Function):
    """rgb_map [B,S,S,3] -> [B,3,is,is], alpha / depth maps [B,S,S] -> [B,is,is]: NHWC -> NCHW, vertical flip and (with
    anti-aliasing) the 2x2 mean of rasterize.py:953-969, forward and backward each as one kernel (csrc/nr_image.hip)."""

    @staticmethod
    def forward(ctx, rgb_map, alpha_map, depth_map, anti_aliasing):
        lib = _lib.load()
        maps = [m.detach().contiguous() if m is not None else None for m in (rgb_map, alpha_map, depth_map)]
        ref = next(m for m in maps if m is not None)
        dev = ref.device
        B, S = ref.shape[0], ref.shape[1]
        if anti_aliasing and S % 2:
            raise ValueError('anti-aliasing needs an even raster size, got %d' % S)
        size = S // 2 if anti_aliasing else S
        outs = [None if maps[0] is None else torch.empty((B, 3, size, size), dtype=torch.float32, device=dev),
                None if maps[1] is None else torch.empty((B, size, size), dtype=torch.float32, device=dev),
                None if maps[2] is None else torch.empty((B, size, size), dtype=torch.float32, device=dev)]
        with torch.cuda.device(dev):
            _lib.check(lib.nr_image_epilogue(_lib.ptr(maps[0]), _lib.ptr(maps[1]), _lib.ptr(maps[2]), _lib.ptr(outs[0]),
                                             _lib.ptr(outs[1]), _lib.ptr(outs[2]), B, S, int(anti_aliasing),
                                             _stream_ptr(dev)), 'nr_image_epilogue')
        ctx.dims = (B, S, bool(anti_aliasing))
        ctx.set_materialize_grads(False)
        return tuple(outs)

    @staticmethod
    def backward(ctx, g_rgb, g_alpha, g_depth):
        lib = _lib.load()
        B, S, aa = ctx.dims
        grads = [g.contiguous() if g is not None else None for g in (g_rgb, g_alpha, g_depth)]
        ref = next((g for g in grads if g is not None), None)
        if ref is None:
            return None, None, None, None
        dev = ref.device
        # an output without gradient keeps `None`, so that the rasterizer's backward skips its terms altogether
        outs = [None if grads[0] is None else torch.empty((B, S, S, 3), dtype=torch.float32, device=dev),
                None if grads[1] is None else torch.empty((B, S, S), dtype=torch.float32, device=dev),
                None if grads[2] is None else torch.empty((B, S, S), dtype=torch.float32, device=dev)]
        with torch.cuda.device(dev):
            _lib.check(lib.nr_image_epilogue_backward(_lib.ptr(grads[0]), _lib.ptr(grads[1]), _lib.ptr(grads[2]),
                                                      _lib.ptr(outs[0]), _lib.ptr(outs[1]), _lib.ptr(outs[2]), B, S,
                                                      int(aa), _stream_ptr(dev)), 'nr_image_epilogue_backward')
        return outs[0], outs[1], outs[2], None


class Rasterize(object):
    """Same constructor and call convention as the reference's `Rasterize` chainer.Function
    (rasterize.py:19-64): `Rasterize(...)(faces[, textures]) -> (rgb, alpha, depth)` in the internal
    layout (rgb [B,S,S,3], alpha/depth [B,S,S], row 0 = bottom), `None` for outputs not requested.
    The intermediate maps of the last call stay on the instance like in the reference (:54-58).

    Two ways in, as on a chainer.Function:
      fn(faces, textures)                          the differentiable call (torch.autograd.Function underneath)
      fn.forward_gpu((faces, textures)), fn.backward_gpu(inputs, grad_outputs)
                                                   the Function protocol itself (rasterize.py:467, :849), no autograd graph:
                                                   what Chainer calls on the reference.  Same kernels, same results; at small
                                                   batches the direct calls save torch's autograd hand-over to its device
                                                   thread, which costs more host time than the launches."""

    def __init__(self, image_size, near, far, eps, background_color, return_rgb=False, return_alpha=False,
                 return_depth=False):
        if not any((return_rgb, return_alpha, return_depth)):
            # nothing to draw (rasterize.py:25-27 raises a bare Exception)
            raise Exception('nothing to draw: none of return_rgb / return_alpha / return_depth is set')
        self.image_size = image_size
        self.near = near
        self.far = far
        self.eps = eps
        self.background_color = background_color
        self.return_rgb = return_rgb
        self.return_alpha = return_alpha
        self.return_depth = return_depth
        self.fix_batch_z = FIX_TEXTURE_BATCH_Z
        self.exact_gradient = EXACT_GRADIENT
        self.graph_replay = GRAPH_REPLAY  # replay the operator from captured HIP graphs (fixed shapes; see GRAPH_REPLAY)
        # [F,3,3] faces of the global batch element 0 when this call renders a shard of a larger batch (SURVEY Q1:
        # the reference samples textures with batch element 0's depths); None = element 0 of this call
        self.faces_z_ref = None
        # buffers of the last call, as on the reference's Function (rasterize.py:39-64).  The reference also keeps
        # face_inv_map and the two sampling maps; the kernels here recompute them instead of storing them, so
        # those three are properties that run the per-stage entry point with the optional pointers when somebody reads them.
        self.faces = self.textures = None
        self.grad_rgb_map = self.grad_alpha_map = self.grad_depth_map = None
        self.rgb_map = self.alpha_map = self.depth_map = None
        self.grad_faces = self.grad_textures = None
        self.face_index_map = None
        self.batch_size = self.num_faces = self.texture_size = None
        self._res = None    # residuals of the last forward (forward_gpu / backward_gpu protocol, lazy maps)
        self._lazy = {}

    # ---- the buffers of the last call (rasterize.py:39-58)
    def _keep(self, r):
        self._res = r
        self._lazy = {}
        self.faces, self.textures = r.faces, r.textures
        self.face_index_map, self.depth_map = r.face_index_map, r.depth_map
        self.rgb_map, self.alpha_map = r.rgb_map, r.alpha_map
        self.batch_size, self.num_faces = r.B, r.F
        self.texture_size = r.ts if self.return_rgb else None

    def _lazy_maps(self, which):
        """face_inv_map [B,S,S,3,3] (rasterize.py:57) or the sampling maps [B,S,S,8] (:47-48) of the last call: not kept by
        the kernels (they recompute them), so reading one runs the per-stage entry point with the optional output pointers."""
        if which in self._lazy:
            return self._lazy[which]
        r = self._res
        if r is None:
            return None
        lib = _lib.load()
        dev = r.faces.device
        B, F, S, ts = r.B, r.F, r.S, r.ts
        f32 = torch.float32
        with _on_device(dev):
            stream = _stream_ptr(dev)
            if which == 'face_inv_map':
                out = torch.empty((B, S, S, 3, 3), dtype=f32, device=dev)
                ws_bytes = lib.nr_forward_workspace_bytes(B, F, S)
                ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=dev)
                fi = torch.empty_like(r.face_index_map)
                wm = torch.empty((B, S, S, 3), dtype=f32, device=dev)
                dm = torch.empty((B, S, S), dtype=f32, device=dev)
                _lib.check(lib.nr_forward_face_index_map(r.faces.data_ptr(), fi.data_ptr(), wm.data_ptr(), dm.data_ptr(),
                                                         out.data_ptr(), None, B, F, S, float(self.near), float(self.far),
                                                         ws.data_ptr(), ws_bytes, stream), 'nr_forward_face_index_map')
                self._lazy['face_inv_map'] = out
            else:
                if r.textures is None or r.light is not None:
                    return None  # no texture sampling in this call (or per-face light colours: no reference counterpart)
                si = torch.empty((B, S, S, 8), dtype=torch.int32, device=dev)
                sw = torch.empty((B, S, S, 8), dtype=f32, device=dev)
                rgb = torch.empty((B, S, S, 3), dtype=f32, device=dev)
                bg = self.background_color if self.background_color is not None else DEFAULT_BACKGROUND_COLOR
                bg = bg.detach().to(device=dev, dtype=f32).contiguous() if torch.is_tensor(bg) else _background_tensor(bg, dev)
                _lib.check(lib.nr_forward_texture_sampling(
                    r.faces.data_ptr(), _lib.ptr(r.z_ref), r.textures.data_ptr(), r.face_index_map.data_ptr(),
                    self.weight_map.data_ptr(), r.depth_map.data_ptr(), rgb.data_ptr(), si.data_ptr(), sw.data_ptr(), bg.data_ptr(),
                    int(tuple(bg.shape) == (B, 3)), None, B, F, S, ts, float(self.eps), r.flags, stream),
                    'nr_forward_texture_sampling')
                self._lazy['sampling_index_map'], self._lazy['sampling_weight_map'] = si, sw
        return self._lazy[which]

    def _get_weight_map(self):
        """weight_map [B,S,S,3] of the last call (rasterize.py:43): the forward stores the weights of covered pixels only; the
        zeros of the others (:479) are filled in here, once, when the attribute is read."""
        if 'weight_map' not in self._lazy:
            r = self._res
            if r is None or r.weight_map is None:
                return None
            self._lazy['weight_map'] = torch.where((r.face_index_map >= 0).unsqueeze(-1), r.weight_map,
                                                   torch.zeros((), dtype=r.weight_map.dtype, device=r.weight_map.device))
        return self._lazy['weight_map']

    def _set_weight_map(self, value):
        self._lazy['weight_map'] = value

    weight_map = property(_get_weight_map, _set_weight_map)
    face_inv_map = property(lambda self: self._lazy_maps('face_inv_map'))
    sampling_index_map = property(lambda self: self._lazy_maps('sampling_index_map'))
    sampling_weight_map = property(lambda self: self._lazy_maps('sampling_weight_map'))

    # ---- the chainer.Function protocol of the reference (rasterize.py:467-513, :849-889)
    def forward_gpu(self, inputs, face_light=None):
        """inputs = (faces,) or (faces, textures) -> (rgb_map, alpha_map, depth_map), `None` for outputs not requested.  Keeps
        the maps on the instance for backward_gpu.  No autograd graph is recorded."""
        faces = inputs[0]
        textures = inputs[1] if len(inputs) > 1 and self.return_rgb else None
        r = _forward_impl(_Config(self), faces, textures, face_light if self.return_rgb else None)
        self._keep(r)
        return (r.rgb_map if self.return_rgb else None, r.alpha_map if self.return_alpha else None,
                r.depth_map if self.return_depth else None)

    forward = forward_gpu

    def backward_gpu(self, inputs, grad_outputs):
        """grad_outputs = (grad_rgb_map, grad_alpha_map, grad_depth_map), `None` = zeros (rasterize.py:858-878) ->
        (grad_faces,) or (grad_faces, grad_textures) like :886-889; with face_light also grad_light (third).  `inputs` is
        accepted for the protocol's sake: the residuals of the last forward_gpu are what is used."""
        r = self._res
        if r is None:
            raise RuntimeError('backward_gpu before forward_gpu')
        g_rgb, g_alpha, g_depth = (tuple(grad_outputs) + (None, None, None))[:3]
        gf, gt, gl = _backward_impl(_Config(self), r, g_rgb, g_alpha, g_depth, self.return_rgb, r.light is not None)
        if gf is None:  # no gradient at all: zeros (:851-853)
            gf = torch.zeros_like(r.faces)
        if not self.return_rgb or len(inputs) < 2:
            return (gf,)
        if gt is None:
            gt = torch.zeros_like(r.textures)
        return (gf, gt) if r.light is None else (gf, gt, gl)

    backward = backward_gpu

    def __call__(self, faces, textures=None, face_light=None):
        """`face_light` (not in the reference; include/nr_hip.h nr_face_light): [B,F,3] colours that multiply the sampled
        colour of each face; `textures` are then the cubes of the original faces ([B,F,...], or [B,F/2,...] when the second
        half of `faces` are fill_back's reversed copies)."""
        cfg = _Config(self)
        if not self.return_rgb:
            textures = face_light = None
        if self.graph_replay and face_light is None:
            entry = _graph_entry(faces, textures, cfg)
            if entry is not None:
                if torch.is_grad_enabled() and (faces.requires_grad or (textures is not None and textures.requires_grad)):
                    # the usual case: every requested output receives a gradient
                    entry.prepare_backward((bool(self.return_rgb), bool(self.return_alpha), bool(self.return_depth),
                                            bool(self.return_rgb and textures is not None and textures.requires_grad)))
                rgb, alpha, depth, fi = _GraphedRasterizeFunction.apply(faces, textures, entry)
                self._res, self._lazy = None, {}
                self.faces, self.textures, self.weight_map = entry.faces, entry.textures, entry.weight_map  # (dense there)
                self.depth_map = depth if depth is not None else entry.depth_map
                self.rgb_map, self.alpha_map, self.face_index_map = rgb, alpha, fi
                self.batch_size, self.num_faces = entry.dims[0], entry.dims[1]
                self.texture_size = entry.dims[3] if self.return_rgb else None
                return rgb, alpha, depth
        rgb, alpha, depth, _ = _RasterizeFunction.apply(faces, textures, cfg, face_light)
        return rgb, alpha, depth


def rasterize_rgbad(
        faces,
        textures=None,
        image_size=DEFAULT_IMAGE_SIZE,
        anti_aliasing=DEFAULT_ANTI_ALIASING,
        near=DEFAULT_NEAR,
        far=DEFAULT_FAR,
        eps=DEFAULT_EPS,
        background_color=DEFAULT_BACKGROUND_COLOR,
        return_rgb=True,
        return_alpha=True,
        return_depth=True,
        faces_z_ref=None,
        graph_replay=None,
        face_light=None,
):
    """RGB, alpha and depth images from faces (and textures for RGB) -- reference rasterize.py:900-977.

    Returns a dict with 'rgb' [B, 3, image_size, image_size], 'alpha' and 'depth' [B, image_size, image_size]
    (None when not requested).  `faces_z_ref` (not in the reference): see Rasterize.faces_z_ref; `face_light` (not in the
    reference): see Rasterize.__call__."""
    inputs = [faces] if textures is None else [faces, textures, face_light]
    size = image_size * 2 if anti_aliasing else image_size  # 2x super-sampling, :945-951
    fn = Rasterize(size, near, far, eps, background_color, return_rgb, return_alpha, return_depth)
    fn.faces_z_ref = faces_z_ref
    if graph_replay is not None:
        fn.graph_replay = bool(graph_replay)
    rgb, alpha, depth = fn(*inputs)
    # transpose & vertical flip (:953-960) and 0.5x down-sampling (:962-969): one HIP kernel per direction
    rgb, alpha, depth = _ImageEpilogue.apply(rgb, alpha, depth, bool(anti_aliasing))
    return {
        'rgb': rgb if return_rgb else None,
        'alpha': alpha if return_alpha else None,
        'depth': depth if return_depth else None,
    }


def rasterize(
        faces,
        textures,
        image_size=DEFAULT_IMAGE_SIZE,
        anti_aliasing=DEFAULT_ANTI_ALIASING,
        near=DEFAULT_NEAR,
        far=DEFAULT_FAR,
        eps=DEFAULT_EPS,
        background_color=DEFAULT_BACKGROUND_COLOR,
        faces_z_ref=None,
        graph_replay=None,
        face_light=None,
):
    """RGB images [B, 3, image_size, image_size] -- reference rasterize.py:980-1008."""
    return rasterize_rgbad(
        faces, textures, image_size, anti_aliasing, near, far, eps, background_color, True, False, False,
        faces_z_ref=faces_z_ref, graph_replay=graph_replay, face_light=face_light)['rgb']


def rasterize_silhouettes(
        faces,
        image_size=DEFAULT_IMAGE_SIZE,
        anti_aliasing=DEFAULT_ANTI_ALIASING,
        near=DEFAULT_NEAR,
        far=DEFAULT_FAR,
        eps=DEFAULT_EPS,
        graph_replay=None,
):
    """Alpha channels [B, image_size, image_size] -- reference rasterize.py:1011-1034."""
    return rasterize_rgbad(faces, None, image_size, anti_aliasing, near, far, eps, None, False, True, False,
                           graph_replay=graph_replay)['alpha']


def rasterize_depth(
        faces,
        image_size=DEFAULT_IMAGE_SIZE,
        anti_aliasing=DEFAULT_ANTI_ALIASING,
        near=DEFAULT_NEAR,
        far=DEFAULT_FAR,
        eps=DEFAULT_EPS,
        graph_replay=None,
):
    """Depth images [B, image_size, image_size] -- reference rasterize.py:1037-1060."""
    return rasterize_rgbad(faces, None, image_size, anti_aliasing, near, far, eps, None, False, False, True,
                           graph_replay=graph_replay)['depth']


def use_unsafe_rasterizer(flag):
    """Kept for API compatibility (rasterize.py:1063-1065, env NEURAL_RENDERER_UNSAFE :15-16).  The reference's "unsafe"
    kernel (K3, :102-236) is a per-face scan conversion with a per-pixel spin lock: same coverage and face indices as
    the safe path, an order-dependent tie rule and weights that differ at the 1e-4 level.  The face-parallel
    rasterizer here already is O(sum of screen boxes) -- what K3 buys the reference -- and deterministic, so both settings
    run it and give bit-identical outputs (tests/test_hip_parity.py::test_unsafe_rasterizer_flag_is_equivalent)."""
    global USE_UNSAFE_IMPLEMENTATION
    USE_UNSAFE_IMPLEMENTATION = flag
