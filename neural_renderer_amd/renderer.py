"""Renderer facade -- reference neural_renderer/renderer.py:8-107 (same attributes, defaults and methods).

On CUDA tensors with the usual parameter types the chain in front of the rasterizer (fill_back, lighting, look_at / look,
perspective, vertices_to_faces) runs as one fused HIP kernel per direction (frontend.py); any other input keeps the
module-by-module path below, which mirrors the reference line by line."""
import math
import os
import sys

import torch

from . import frontend
from .lighting import lighting
from .look import look
from .look_at import look_at
from .perspective import perspective
from .rasterize import rasterize, rasterize_depth, rasterize_silhouettes
from .vertices_to_faces import vertices_to_faces

# Renderer.face_light default (see the attribute): NR_FACE_LIGHT = 0 | 1 | auto
FACE_LIGHT = {'0': False, '1': True}.get(os.environ.get('NR_FACE_LIGHT', 'auto'))


class Renderer(object):
    def __init__(self):
        # rendering
        self.image_size = 256
        self.anti_aliasing = True
        self.background_color = [0, 0, 0]
        self.fill_back = True

        # camera
        self.perspective = True
        self.viewing_angle = 30
        self.eye = [0, 0, -(1. / math.tan(math.radians(self.viewing_angle)) + 1)]
        self.camera_mode = 'look_at'
        self.camera_direction = [0, 0, 1]
        self.near = 0.1
        self.far = 100

        # light
        self.light_intensity_ambient = 0.5
        self.light_intensity_directional = 0.5
        self.light_color_ambient = [1, 1, 1]  # white
        self.light_color_directional = [1, 1, 1]  # white
        self.light_direction = [0, 1, 0]  # up-to-down

        # rasterization
        self.rasterizer_eps = 1e-3

        # not in the reference: which implementation of the chain in front of the rasterizer the calls took -- 'fused'
        # (one HIP kernel per direction, frontend.py) or 'torch' (module by module).  A benchmark asserts on this so that it
        # cannot fall onto the ~160-launch path unnoticed.
        self.last_frontend = None
        self.frontend_calls = {'fused': 0, 'torch': 0}
        # [F,3,3] faces of the global batch element 0 when this renderer draws a shard of a larger batch
        self.faces_z_ref = None
        # not in the reference: replay the rasterizer from captured HIP graphs (fixed shapes; None = the module default,
        # neural_renderer_amd.use_graph_replay / NR_GRAPH_REPLAY; see rasterize.py)
        self.graph_replay = None
        # not in the reference: render() hands the rasterizer the ORIGINAL textures plus one light colour per face instead of
        # lit, fill_back-duplicated textures (include/nr_hip.h: nr_face_light).  Same images up to the rounding
        # order of the light product (measured 3e-7 of the largest colour, tests/test_face_light_gpu.py), a fraction of the
        # memory traffic on textured meshes (scripts/face_light_timing.py: config 4's shape 2.20 -> 1.12 ms per render +
        # backward at texture_size 4, 16.3 -> 1.85 ms at 8; neutral at 2).  True / False, or None = when it pays
        # (texture_size >= 3).  Needs the fused front-end and no graph replay; otherwise, and with False,
        # the lit-texture path runs.  Default: NR_FACE_LIGHT (auto).
        self.face_light = FACE_LIGHT

    def _project(self, vertices, faces):
        """camera + perspective + gather (renderer.py:40-51, :60-71, :92-103)."""
        if self.camera_mode == 'look_at':
            vertices = look_at(vertices, self.eye)
        elif self.camera_mode == 'look':
            vertices = look(vertices, self.eye, self.camera_direction)
        if self.perspective:
            vertices = perspective(vertices, angle=self.viewing_angle)
        return vertices_to_faces(vertices, faces)

    def _frontend_torch(self, vertices, faces, textures=None):
        """Everything in front of the rasterizer, module by module as in the reference (renderer.py:37-51, :77-103)."""
        if self.fill_back:  # renderer.py:37-38, :77-79
            faces = torch.cat((faces, torch.flip(faces, dims=[2])), dim=1).detach()
            if textures is not None:
                textures = torch.cat((textures, textures.permute(0, 1, 4, 3, 2, 5)), dim=1)
        if textures is not None:  # lighting in world space (renderer.py:82-90)
            faces_lighting = vertices_to_faces(vertices, faces)
            textures = lighting(
                faces_lighting,
                textures,
                self.light_intensity_ambient,
                self.light_intensity_directional,
                self.light_color_ambient,
                self.light_color_directional,
                self.light_direction)
        return self._project(vertices, faces), textures

    def _frontend(self, vertices, faces, textures=None):
        """-> (faces [B,F,3,3], lit textures | None): the fused HIP front-end when the call fits it, else torch."""
        fused = frontend.fusable(self, vertices, faces, textures)
        self.last_frontend = 'fused' if fused else 'torch'
        self.frontend_calls[self.last_frontend] += 1
        if fused:
            return frontend.project_and_light(self, vertices, faces, textures)
        return self._frontend_torch(vertices, faces, textures)

    def render_silhouettes(self, vertices, faces):
        faces, _ = self._frontend(vertices, faces)
        # NB: near / far / rasterizer_eps are NOT forwarded here (renderer.py:52)
        return rasterize_silhouettes(faces, self.image_size, self.anti_aliasing, graph_replay=self.graph_replay)

    def render_depth(self, vertices, faces):
        faces, _ = self._frontend(vertices, faces)
        return rasterize_depth(faces, self.image_size, self.anti_aliasing, graph_replay=self.graph_replay)  # renderer.py:72 (Q2)

    def _use_face_light(self, vertices, faces, textures):
        if self.face_light is False or not (torch.is_tensor(textures) and textures.dim() == 6):
            return False
        ts = textures.shape[2]
        if self.face_light is None and ts < 3:
            return False
        # (the package attribute `rasterize` is the function; the module of that name holds the switch)
        replay = self.graph_replay if self.graph_replay is not None else sys.modules[rasterize.__module__].GRAPH_REPLAY
        return not replay and frontend.fusable(self, vertices, faces, textures)

    def render(self, vertices, faces, textures):
        if self._use_face_light(vertices, faces, textures):
            self.last_frontend = 'fused'
            self.frontend_calls['fused'] += 1
            faces, light = frontend.project_and_light_colors(self, vertices, faces)
            return rasterize(
                faces, textures, self.image_size, self.anti_aliasing, self.near, self.far, self.rasterizer_eps,
                self.background_color, faces_z_ref=self.faces_z_ref, face_light=light)
        faces, textures = self._frontend(vertices, faces, textures)
        return rasterize(
            faces, textures, self.image_size, self.anti_aliasing, self.near, self.far, self.rasterizer_eps,
            self.background_color, faces_z_ref=self.faces_z_ref, graph_replay=self.graph_replay)
