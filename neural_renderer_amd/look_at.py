""""Look at" transformation of vertices -- reference neural_renderer/look_at.py:7-46."""
import torch

from ._util import normalize, as_tensor_like
from .cross import cross


def look_at(vertices, eye, at=None, up=None):
    assert vertices.dim() == 3
    batch_size = vertices.shape[0]
    at = as_tensor_like([0, 0, 0] if at is None else at, vertices)
    up = as_tensor_like([0, 1, 0] if up is None else up, vertices)
    eye = as_tensor_like(eye, vertices)  # list / tuple / array / (learnable) tensor, look_at.py:20-21
    if eye.dim() == 1:
        eye = eye[None, :].repeat(batch_size, 1)
    if at.dim() == 1:
        at = at[None, :].repeat(batch_size, 1)
    if up.dim() == 1:
        up = up[None, :].repeat(batch_size, 1)

    # create new axes
    z_axis = normalize(at - eye)
    x_axis = normalize(cross(up, z_axis))
    y_axis = normalize(cross(z_axis, x_axis))

    # rotation matrix: [bs, 3, 3]
    r = torch.cat((x_axis[:, None, :], y_axis[:, None, :], z_axis[:, None, :]), dim=1)
    if r.shape[0] != vertices.shape[0]:
        r = r.expand(vertices.shape[0], 3, 3)

    # apply: [bs, nv, 3] -> [bs, nv, 3]
    if vertices.shape != eye.shape:
        eye = eye[:, None, :].expand_as(vertices)
    vertices = vertices - eye
    vertices = torch.matmul(vertices, r.transpose(1, 2))
    return vertices
