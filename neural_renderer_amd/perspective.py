"""Perspective division -- reference neural_renderer/perspective.py:5-19 (note pi = 3.1416, :10)."""
import torch


def perspective(vertices, angle=30.):
    assert vertices.dim() == 3
    if isinstance(angle, float) or isinstance(angle, int):
        angle = torch.tensor(angle, dtype=torch.float32, device=vertices.device)
    angle = angle / 180. * 3.1416
    angle = angle[None].expand(vertices.shape[0])

    width = torch.tan(angle)
    width = width[:, None].expand(vertices.shape[:2])
    z = vertices[:, :, 2]
    x = vertices[:, :, 0] / z / width
    y = vertices[:, :, 1] / z / width
    vertices = torch.cat((x[:, :, None], y[:, :, None], z[:, :, None]), dim=2)
    return vertices
