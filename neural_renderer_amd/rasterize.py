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
import os
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

# K6 (backward_pixel_map) numerics.  Default: float terms through fused multiply-adds and the hardware reciprocal, measured <= 4.7e-5 from the
# reference's terms summed exactly (tests bound it by the north star's 1e-4).  NR_EXACT_GRADIENT=1 (read once, here) or the
# `exact_gradient` attribute of a Rasterize instance: every term with the reference's own arithmetic, <= 2e-6, ~1.5x the K6 time.
EXACT_GRADIENT = bool(int(os.environ.get('NR_EXACT_GRADIENT', '0')))


def _stream_ptr(device):
    return torch.cuda.current_stream(device).cuda_stream


_BG_CACHE = {}


def _background_tensor(bg, device):
    """Device copy of a host background colour, cached (a pageable H2D copy per call would serialise the stream)."""
    arr = np.asarray(bg, dtype=np.float32)
    key = (str(device), arr.shape, arr.tobytes())
    t = _BG_CACHE.get(key)
    if t is None:
        if len(_BG_CACHE) > 64:
            _BG_CACHE.clear()
        t = _BG_CACHE[key] = torch.as_tensor(arr, device=device)
    return t


# Graph replay (not in the reference): for fixed shapes the forward and the backward of the operator are each captured once
# in a HIP graph and replayed from then on -- at small batches the ~12 launches of a step take longer to ISSUE than to run
# (BASELINE config 2, 16 views: 0.27 ms eager, 0.18 ms replayed).  Off by default; `use_graph_replay(True)`, the
# NR_GRAPH_REPLAY=1 environment variable (read once, here) or the `graph_replay` attribute of a Rasterize / Renderer
# instance switch it on.  Semantics stay those of the eager operator: inputs are copied into the graphs' fixed buffers and
# results are returned as copies, so callers may keep them; what replay cannot offer is two forwards of the same shapes in
# flight before the first one's backward (the residual maps live in the fixed buffers) -- that raises.  Measured (config 2):
# 0.25 ms eager -> 0.21 ms with replay; a step captured as a whole by the caller (neural_renderer_amd.graph.capture: no
# copies, loss and optimizer inside) 0.17 ms -- prefer that where the whole step is fixed.  Known limitation (ROCm 7.2 /
# torch 2.10): a whole-step capture started AFTER this mode has run in the same process crashes; use one or the other.
GRAPH_REPLAY = bool(int(os.environ.get('NR_GRAPH_REPLAY', '0')))


def use_graph_replay(flag):
    global GRAPH_REPLAY
    GRAPH_REPLAY = bool(flag)


_ZBUF_CACHE = {}


def _forward_workspace(lib, dev, stream, B, F, S):
    """The forward's scratch (z-buffer + queues) and the flags that go with it.  Kept per (device, stream, sizes) and reused
    with a falling epoch number (include/nr_hip.h: NR_FLAG_ZBUF_EPOCH), so that the library neither fills the z-buffer
    before a call nor cleans it afterwards; refilled with 0xff every 255 calls.  While a HIP graph is being captured the
    epoch would be frozen into the graph, so capture takes a throw-away workspace and the filling path."""
    ws_bytes = lib.nr_forward_workspace_bytes(B, F, S)
    if ws_bytes == 0:
        raise ValueError('unsupported sizes B=%d F=%d S=%d' % (B, F, S))
    if F >= (1 << 24) or torch.cuda.is_current_stream_capturing():
        return torch.empty((ws_bytes,), dtype=torch.uint8, device=dev), ws_bytes, 0
    key = (dev.index, int(stream), B, F, S)
    ent = _ZBUF_CACHE.get(key)
    if ent is None:
        if len(_ZBUF_CACHE) >= 8:  # a handful of shapes at a time: drop the oldest
            _ZBUF_CACHE.pop(next(iter(_ZBUF_CACHE)))
        ent = _ZBUF_CACHE[key] = [torch.empty((ws_bytes,), dtype=torch.uint8, device=dev), -1]
    if ent[1] < 0:
        ent[0].fill_(255)
        ent[1] = 254
    epoch = ent[1]
    ent[1] -= 1
    return ent[0], ws_bytes, _lib.NR_FLAG_ZBUF_EPOCH | (epoch << 8)


class _RasterizeFunction(torch.autograd.Function):
    """forward(ctx, faces, textures, cfg, light) -> (rgb_map [B,S,S,3] | None, alpha_map [B,S,S] | None,
    depth_map [B,S,S] | None); backward(ctx, g_rgb, g_alpha, g_depth) -> (grad_faces, grad_textures, None, grad_light).
    `light` [B,F,3] (or None): per-face light colours, textures are then the original cubes [B,Nf,...] with F = Nf or 2 Nf
    (include/nr_hip.h: nr_face_light)."""

    @staticmethod
    def forward(ctx, faces, textures, cfg, light=None):
        lib = _lib.load()
        if not faces.is_cuda:
            raise NotImplementedError('neural_renderer_amd has no CPU rasterizer (neither has the reference: '
                                      'rasterize.py:893-897)')
        if faces.dtype != torch.float32 or faces.dim() != 4 or tuple(faces.shape[2:]) != (3, 3):
            raise ValueError('faces must be float32 [batch size, num of faces, 3, 3], got %s %s'
                             % (faces.dtype, tuple(faces.shape)))
        return_rgb, return_alpha, return_depth = cfg['return_rgb'], cfg['return_alpha'], cfg['return_depth']
        dev = faces.device
        faces_c = faces.detach().contiguous()  # rasterize.py:470
        B, F = faces_c.shape[:2]
        S = int(cfg['image_size'])
        ts = 0
        textures_c = light_c = None
        Nf = F
        if not return_rgb:
            light = None
        if return_rgb:
            if textures is None:
                raise ValueError('textures are required when return_rgb is set')
            if light is not None:
                if light.dtype != torch.float32 or tuple(light.shape) != (B, F, 3):
                    raise ValueError('face_light must be float32 [batch size, num of faces, 3], got %s %s'
                                     % (light.dtype, tuple(light.shape)))
                if textures.dim() == 6 and textures.shape[1] * 2 == F:
                    Nf = F // 2  # fill_back: face Nf + f is the reversed copy of face f
                light_c = light.detach().contiguous()
            if (textures.dtype != torch.float32 or textures.dim() != 6 or textures.shape[0] != B or
                    textures.shape[1] != Nf or textures.shape[2] < 2 or textures.shape[2] != textures.shape[3] or
                    textures.shape[3] != textures.shape[4] or textures.shape[5] != 3):
                raise ValueError('textures must be float32 [batch size, num of faces, ts, ts, ts, 3] with ts >= 2, '
                                 'got %s %s' % (textures.dtype, tuple(textures.shape)))  # rasterize.py:78-90
            textures_c = textures.detach().contiguous()  # :473
            ts = int(textures_c.shape[2])

        with torch.cuda.device(dev):
            stream = _stream_ptr(dev)
            need_wd = return_rgb or return_depth
            face_index_map = torch.empty((B, S, S), dtype=torch.int32, device=dev)
            weight_map = torch.empty((B, S, S, 3), dtype=torch.float32, device=dev) if need_wd else None
            depth_map = torch.empty((B, S, S), dtype=torch.float32, device=dev) if need_wd else None
            workspace, ws_bytes, ws_flags = _forward_workspace(lib, dev, stream, B, F, S)
            rgb_map = alpha_map = background = None
            bg_per_batch = 0
            if return_rgb:
                rgb_map = torch.empty((B, S, S, 3), dtype=torch.float32, device=dev)
                bg = cfg['background_color']
                if torch.is_tensor(bg):
                    background = bg.detach().to(device=dev, dtype=torch.float32).contiguous()
                else:
                    background = _background_tensor(bg, dev)
                if tuple(background.shape) == (B, 3):
                    bg_per_batch = 1  # rasterize.py:464-465
                elif tuple(background.shape) != (3,):
                    raise ValueError('background_color must have shape (3,) or (batch size, 3)')
            if return_alpha:
                alpha_map = torch.empty((B, S, S), dtype=torch.float32, device=dev)
            flags = _lib.NR_FLAG_FIX_TEXTURE_BATCH_Z if cfg['fix_batch_z'] else 0
            if cfg['exact_gradient']:
                flags |= _lib.NR_FLAG_EXACT_GRADIENT
            # batch element 0 of the GLOBAL batch when this call holds a shard of it (include/nr_hip.h)
            z_ref = cfg.get('faces_z_ref')
            if z_ref is not None:
                z_ref = z_ref.detach().to(device=dev, dtype=torch.float32).contiguous()
                if tuple(z_ref.shape) != (F, 3, 3):
                    raise ValueError('faces_z_ref must have shape (num of faces, 3, 3), got %s' % (tuple(z_ref.shape),))
            # per-face "owns a pixel" flags: a residual the backward starts from (K6's lists; the depth-only gather skips the
            # faces without a pixel)
            visible = torch.empty((B, F), dtype=torch.uint8, device=dev)
            # visibility + shading behind one call (rasterize.py:499-502)
            lit = None
            if light_c is not None:
                lit = _lib.FaceLight(light_c.data_ptr(), Nf, None, None)
            _lib.check(lib.nr_forward_rasterize_lit(
                lit, faces_c.data_ptr(), _lib.ptr(z_ref), _lib.ptr(textures_c), face_index_map.data_ptr(),
                _lib.ptr(weight_map), _lib.ptr(depth_map), _lib.ptr(rgb_map), _lib.ptr(alpha_map), _lib.ptr(visible),
                _lib.ptr(background), bg_per_batch, B, F, S, ts, float(cfg['near']), float(cfg['far']),
                float(cfg['eps']), flags | ws_flags, workspace.data_ptr(), ws_bytes, stream), 'nr_forward_rasterize')

        ctx.cfg = dict(cfg, keep=None, B=B, F=F, S=S, ts=ts, flags=flags, Nf=Nf)
        ctx.lit = (light_c, textures_c) if light_c is not None else None
        keep = cfg.get('keep')
        if keep is not None:  # the maps the reference leaves on the Function instance (rasterize.py:39-58); no copies
            keep.update(faces=faces_c, textures=textures_c, face_index_map=face_index_map, weight_map=weight_map,
                        depth_map=depth_map, rgb_map=rgb_map, alpha_map=alpha_map, batch_size=B, num_faces=F,
                        texture_size=ts if return_rgb else None)
        ctx.z_ref = z_ref
        ctx.visible = visible
        ctx.set_materialize_grads(False)  # an unused output arrives as `None` in backward and its terms are skipped
        # residuals (the reference keeps them on `self`, rasterize.py:39-58); outputs among them go through
        # save_for_backward so that in-place edits by the caller are detected
        ctx.save_for_backward(faces_c, face_index_map, weight_map, depth_map, rgb_map, alpha_map)
        ctx.mark_non_differentiable(face_index_map)
        outs = (rgb_map if return_rgb else None, alpha_map if return_alpha else None,
                depth_map if return_depth else None, face_index_map)
        return outs

    @staticmethod
    def backward(ctx, g_rgb, g_alpha, g_depth, _g_fi):
        lib = _lib.load()
        cfg = ctx.cfg
        B, F, S, ts = cfg['B'], cfg['F'], cfg['S'], cfg['ts']
        faces_c, face_index_map, weight_map, depth_map, rgb_map, alpha_map = ctx.saved_tensors
        dev = faces_c.device
        # None gradients are zeros (rasterize.py:858-878); a zero gradient adds exactly 0 to every
        # `diff_grad`, so the corresponding term is skipped instead of being multiplied out.
        use_rgb = cfg['return_rgb'] and g_rgb is not None
        use_alpha = cfg['return_alpha'] and g_alpha is not None
        use_depth = cfg['return_depth'] and g_depth is not None
        if not (use_rgb or use_alpha or use_depth):
            return None, None, None, None
        grad_light = None
        with torch.cuda.device(dev):
            stream = _stream_ptr(dev)
            if use_rgb:
                g_rgb = g_rgb.contiguous()
            if use_alpha:
                g_alpha = g_alpha.contiguous()
            if use_depth:
                g_depth = g_depth.contiguous()
            grad_faces = torch.empty_like(faces_c)  # stored by the library (zeros when neither rgb nor alpha)
            grad_textures = None
            lit = None
            if use_rgb and ctx.lit is not None:
                if ctx.needs_input_grad[1] or ctx.needs_input_grad[3]:
                    # one gather produces both (the colours' gradient is a by-product of the texel sums)
                    grad_textures = torch.empty((B, cfg['Nf'], ts, ts, ts, 3), dtype=torch.float32, device=dev)
                    if ctx.needs_input_grad[3]:
                        grad_light = torch.empty((B, F, 3), dtype=torch.float32, device=dev)
                lit = _lib.FaceLight(ctx.lit[0].data_ptr(), cfg['Nf'], ctx.lit[1].data_ptr(), _lib.ptr(grad_light))
            elif use_rgb and ctx.needs_input_grad[1]:
                grad_textures = torch.empty((B, F, ts, ts, ts, 3), dtype=torch.float32, device=dev)
            ws_bytes = lib.nr_backward_workspace_bytes(B, F, S, int(use_rgb), int(use_alpha))
            workspace = torch.empty((max(ws_bytes, 1),), dtype=torch.uint8, device=dev)
            # K6 -> K7 -> K8 (rasterize.py:881-883) behind one call
            _lib.check(lib.nr_backward_rasterize_lit(
                lit, faces_c.data_ptr(), _lib.ptr(ctx.z_ref), face_index_map.data_ptr(), _lib.ptr(weight_map),
                _lib.ptr(depth_map), _lib.ptr(rgb_map) if use_rgb else None, _lib.ptr(alpha_map) if use_alpha else None,
                _lib.ptr(g_rgb) if use_rgb else None, _lib.ptr(g_alpha) if use_alpha else None,
                _lib.ptr(g_depth) if use_depth else None, grad_faces.data_ptr(), _lib.ptr(grad_textures),
                B, F, S, ts, float(cfg['eps']), cfg['flags'], _lib.ptr(ctx.visible), workspace.data_ptr(), ws_bytes,
                stream), 'nr_backward_rasterize')
        owner = cfg['owner']() if cfg.get('owner') is not None else None
        if owner is not None:  # rasterize.py:41-51: the gradient buffers stay readable on the instance
            owner.grad_rgb_map, owner.grad_alpha_map, owner.grad_depth_map = g_rgb, g_alpha, g_depth
            # (aliases, not the returned tensors themselves: a second reference to a returned gradient makes autograd's
            # AccumulateGrad clone it instead of adopting it -- two device copies, 41 MB per step at the headline size)
            owner.grad_faces = grad_faces.detach()
            owner.grad_textures = grad_textures.detach() if grad_textures is not None else None
        if not ctx.needs_input_grad[1]:
            grad_textures = None
        return grad_faces, grad_textures, None, grad_light


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
        rgb, alpha, depth = cfg['return_rgb'], cfg['return_alpha'], cfg['return_depth']
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
        self.flags = (_lib.NR_FLAG_FIX_TEXTURE_BATCH_Z if cfg['fix_batch_z'] else 0) | \
                     (_lib.NR_FLAG_EXACT_GRADIENT if cfg['exact_gradient'] else 0)
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
            self.visible.data_ptr(), _lib.ptr(self.background), self.bg_per_batch, B, F, S, ts, float(cfg['near']),
            float(cfg['far']), float(cfg['eps']), self.flags, self.fwd_ws.data_ptr(), self.fwd_ws.numel(),
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
                float(self.cfg['eps']), self.flags, self.visible.data_ptr(), buf['ws'].data_ptr(), ws_bytes,
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
        return (entry.rgb_map.clone() if cfg['return_rgb'] else None, entry.alpha_map.clone() if cfg['return_alpha'] else None,
                entry.depth_map.clone() if cfg['return_depth'] else None, fi)

    @staticmethod
    def backward(ctx, g_rgb, g_alpha, g_depth, _g_fi):
        entry, cfg = ctx.entry, ctx.entry.cfg
        if entry.generation != ctx.generation:
            raise RuntimeError('graph replay: a later forward of the same shapes has replaced the residual maps of this call; '
                               'run backward before the next forward of these shapes, or switch graph replay off')
        use_rgb = cfg['return_rgb'] and g_rgb is not None
        use_alpha = cfg['return_alpha'] and g_alpha is not None
        use_depth = cfg['return_depth'] and g_depth is not None
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
    S = int(cfg['image_size'])
    ts = 0
    bg, bg_per_batch, bg_key = None, 0, None
    if cfg['return_rgb']:
        if textures is None or textures.dim() != 6 or textures.dtype != torch.float32 or tuple(textures.shape[:2]) != (B, F):
            return None
        ts = int(textures.shape[2])
        b = cfg['background_color']
        if torch.is_tensor(b):
            return None  # a device-side background may change between calls: eager path
        arr = np.asarray(b, dtype=np.float32)
        if arr.shape == (B, 3):
            bg_per_batch = 1
        elif arr.shape != (3,):
            return None
        bg, bg_key = _background_tensor(b, dev), arr.tobytes()
    if cfg.get('faces_z_ref') is not None:
        return None  # sharded batches hand a reference view over per call: eager path
    key = (dev.index, B, F, S, ts, cfg['return_rgb'], cfg['return_alpha'], cfg['return_depth'], float(cfg['near']),
           float(cfg['far']), float(cfg['eps']), bg_key, bg_per_batch, cfg['fix_batch_z'], cfg['exact_gradient'])
    entry = _GRAPH_CACHE.get(key)
    if entry is None:
        if len(_GRAPH_CACHE) >= 8:
            _GRAPH_CACHE.pop(next(iter(_GRAPH_CACHE)))
        entry = _GRAPH_CACHE[key] = _GraphEntry(_lib.load(), dev, dict(cfg), B, F, S, ts, bg, bg_per_batch, False)
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
    The intermediate maps of the last call stay on the instance like in the reference (:54-58)."""

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
        # face_inv_map and the two sampling maps; here they are recomputed inside the backward kernels instead of being
        # stored, so those three attributes stay None.
        self.faces = self.textures = None
        self.grad_rgb_map = self.grad_alpha_map = self.grad_depth_map = None
        self.rgb_map = self.alpha_map = self.depth_map = None
        self.grad_faces = self.grad_textures = None
        self.face_index_map = self.weight_map = None
        self.face_inv_map = self.sampling_index_map = self.sampling_weight_map = None
        self.batch_size = self.num_faces = self.texture_size = None

    def __call__(self, faces, textures=None, face_light=None):
        """`face_light` (not in the reference; include/nr_hip.h nr_face_light): [B,F,3] colours that multiply the sampled
        colour of each face; `textures` are then the cubes of the original faces ([B,F,...], or [B,F/2,...] when the second
        half of `faces` are fill_back's reversed copies)."""
        keep = {}
        cfg = dict(keep=keep, owner=weakref.ref(self), image_size=self.image_size, near=self.near, far=self.far, eps=self.eps,
                   background_color=self.background_color if self.background_color is not None
                   else DEFAULT_BACKGROUND_COLOR,
                   return_rgb=bool(self.return_rgb), return_alpha=bool(self.return_alpha),
                   return_depth=bool(self.return_depth), fix_batch_z=bool(self.fix_batch_z),
                   exact_gradient=bool(self.exact_gradient), faces_z_ref=self.faces_z_ref)
        if not self.return_rgb:
            textures = face_light = None
        entry = _graph_entry(faces, textures, cfg) if self.graph_replay and face_light is None else None
        if entry is not None and torch.is_grad_enabled() and (faces.requires_grad or (textures is not None and textures.requires_grad)):
            # the usual case: every requested output receives a gradient
            entry.prepare_backward((bool(self.return_rgb), bool(self.return_alpha), bool(self.return_depth),
                                    bool(self.return_rgb and textures is not None and textures.requires_grad)))
        if entry is not None:
            rgb, alpha, depth, fi = _GraphedRasterizeFunction.apply(faces, textures, entry)
            keep.update(faces=entry.faces, textures=entry.textures, weight_map=entry.weight_map, depth_map=depth if depth is not None else entry.depth_map,
                        rgb_map=rgb, alpha_map=alpha, batch_size=entry.dims[0], num_faces=entry.dims[1],
                        texture_size=entry.dims[3] if self.return_rgb else None)
        else:
            rgb, alpha, depth, fi = _RasterizeFunction.apply(faces, textures, cfg, face_light)
        for k, v in keep.items():
            setattr(self, k, v)
        self.face_index_map = fi
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
