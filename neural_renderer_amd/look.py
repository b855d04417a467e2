""""Look" transformation of vertices (camera at `eye` looking along `direction`) -- reference
neural_renderer/look.py:7-45.  The reference broadcasts the rotation to `vertices.shape` (look.py:36),
which only works for 3-vertex inputs; here it is broadcast to [batch, 3, 3] as the
code evidently intends."""
import torch

from ._util import normalize, as_tensor_like
from .cross import cross


def look(vertices, eye, direction=None, up=None):
    assert vertices.dim() == 3
    direction = as_tensor_like([0, 0, 1] if direction is None else direction, vertices)
    up = as_tensor_like([0, 1, 0] if up is None else up, vertices)
    eye = as_tensor_like(eye, vertices)
    if eye.dim() == 1:
        eye = eye[None, :]
    if direction.dim() == 1:
        direction = direction[None, :]
    if up.dim() == 1:
        up = up[None, :]

    z_axis = normalize(direction)
    x_axis = normalize(cross(up.expand_as(z_axis), z_axis))
    y_axis = normalize(cross(z_axis, x_axis))

    r = torch.cat((x_axis[:, None, :], y_axis[:, None, :], z_axis[:, None, :]), dim=1)
    if r.shape[0] != vertices.shape[0]:
        r = r.expand(vertices.shape[0], 3, 3)

    if vertices.shape != eye.shape:
        eye = eye[:, None, :].expand_as(vertices)
    vertices = vertices - eye
    vertices = torch.matmul(vertices, r.transpose(1, 2))
    return vertices
