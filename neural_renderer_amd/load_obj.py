"""Wavefront .obj loader (vertices + faces) -- reference neural_renderer/load_obj.py:147-197.
Texture baking (`load_texture=True`, load_obj.py:25-144) is a "next" row and raises."""
import numpy as np


def load_obj(filename_obj, normalization=True, texture_size=4, load_texture=False):
    """Returns (vertices [Nv,3] float32, faces [Nf,3] int32). Polygons are fan-triangulated (:167-175);
    with `normalization` the mesh is scaled into the unit cube centred at zero (:188-192)."""
    if load_texture:
        raise NotImplementedError('load_obj(load_texture=True) is not part of the rasterizer hot path yet')
    vertices = []
    faces = []
    with open(filename_obj) as f:
        for line in f:
            t = line.split()
            if len(t) == 0:
                continue
            if t[0] == 'v':
                vertices.append([float(v) for v in t[1:4]])
            elif t[0] == 'f':
                vs = t[1:]
                v0 = int(vs[0].split('/')[0])
                for i in range(len(vs) - 2):
                    v1 = int(vs[i + 1].split('/')[0])
                    v2 = int(vs[i + 2].split('/')[0])
                    faces.append((v0, v1, v2))
    vertices = np.vstack(vertices).astype('float32')
    faces = np.vstack(faces).astype('int32') - 1

    if normalization:
        vertices -= vertices.min(0)[None, :]
        vertices /= np.abs(vertices).max()
        vertices *= 2
        vertices -= vertices.max(0)[None, :] / 2
    return vertices, faces
