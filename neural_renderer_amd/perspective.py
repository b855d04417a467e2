"""Perspective division -- reference neural_renderer/perspective.py:5-19.

x and y of every camera-space vertex are divided by z * tan(angle); z is kept.  The angle is given in degrees and
converted with the reference's own constant pi = 3.1416 (perspective.py:10), in float32, so that projected coordinates
agree with the reference (and with the fused HIP front-end, which receives the same float32 tangent)."""
import torch

_PI_REFERENCE = 3.1416  # sic: not math.pi


def perspective(vertices, angle=30.):
    assert vertices.dim() == 3
    if not torch.is_tensor(angle):
        angle = torch.tensor(float(angle), dtype=torch.float32, device=vertices.device)
    tangent = torch.tan(angle / 180. * _PI_REFERENCE)          # 0-d, or one angle per batch element
    tangent = tangent.reshape(-1, 1) if tangent.dim() else tangent
    depth = vertices[..., 2]
    return torch.stack((vertices[..., 0] / depth / tangent, vertices[..., 1] / depth / tangent, depth), dim=2)
