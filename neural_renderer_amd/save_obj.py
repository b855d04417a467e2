"""Wavefront .obj writer (geometry only) -- reference neural_renderer/save_obj.py:150-191 without the
texture-atlas export (out of scope)."""
import numpy as np


def save_obj(filename, vertices, faces, textures=None):
    assert vertices.ndim == 2 and faces.ndim == 2
    if textures is not None:
        raise NotImplementedError('texture atlas export is out of scope of the rasterizer hot path')
    vertices = np.asarray(vertices)
    faces = np.asarray(faces)
    with open(filename, 'w') as f:
        for v in vertices:
            f.write('v %.9g %.9g %.9g\n' % (v[0], v[1], v[2]))
        f.write('\n')
        for face in faces:
            f.write('f %d %d %d\n' % (face[0] + 1, face[1] + 1, face[2] + 1))
