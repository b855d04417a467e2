"""Per-face ambient + directional (Lambert) light multiplied into the textures -- reference
neural_renderer/lighting.py:8-51."""
import torch

from ._util import normalize, as_tensor_like
from .cross import cross


def lighting(
        faces, textures, intensity_ambient=0.5, intensity_directional=0.5, color_ambient=(1, 1, 1),
        color_directional=(1, 1, 1), direction=(0, 1, 0)):
    bs, nf = faces.shape[:2]

    color_ambient = as_tensor_like(color_ambient, faces)
    color_directional = as_tensor_like(color_directional, faces)
    direction = as_tensor_like(direction, faces)
    if color_ambient.dim() == 1:
        color_ambient = color_ambient[None, :].expand(bs, 3)
    if color_directional.dim() == 1:
        color_directional = color_directional[None, :].expand(bs, 3)
    if direction.dim() == 1:
        direction = direction[None, :].expand(bs, 3)

    light = torch.zeros((bs, nf, 3), dtype=torch.float32, device=faces.device)

    # ambient light
    if intensity_ambient != 0:
        light = light + intensity_ambient * color_ambient[:, None, :].expand_as(light)

    # directional light
    if intensity_directional != 0:
        f = faces.reshape((bs * nf, 3, 3))
        v10 = f[:, 0] - f[:, 1]
        v12 = f[:, 2] - f[:, 1]
        normals = normalize(cross(v10, v12))
        normals = normals.reshape((bs, nf, 3))

        if direction.dim() == 2:
            direction = direction[:, None, :].expand_as(normals)
        cos = torch.relu(torch.sum(normals * direction, dim=2))
        light = light + intensity_directional * (color_directional[:, None, :] * cos[:, :, None])

    # apply
    light = light[:, :, None, None, None, :].expand_as(textures)
    textures = textures * light
    return textures
