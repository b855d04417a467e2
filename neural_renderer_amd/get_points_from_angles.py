"""Camera position from distance / elevation / azimuth -- reference get_points_from_angles.py:6-24."""
import math

import torch


def get_points_from_angles(distance, elevation, azimuth, degrees=True):
    if isinstance(distance, float) or isinstance(distance, int):
        if degrees:
            elevation = math.radians(elevation)
            azimuth = math.radians(azimuth)
        return (
            distance * math.cos(elevation) * math.sin(azimuth),
            distance * math.sin(elevation),
            -distance * math.cos(elevation) * math.cos(azimuth))
    else:
        distance = torch.as_tensor(distance)
        elevation = torch.as_tensor(elevation, dtype=distance.dtype, device=distance.device)
        azimuth = torch.as_tensor(azimuth, dtype=distance.dtype, device=distance.device)
        if degrees:
            elevation = torch.deg2rad(elevation)
            azimuth = torch.deg2rad(azimuth)
        return torch.stack([
            distance * torch.cos(elevation) * torch.sin(azimuth),
            distance * torch.sin(elevation),
            -distance * torch.cos(elevation) * torch.cos(azimuth),
        ]).transpose(0, -1) if distance.dim() > 0 else torch.stack([
            distance * torch.cos(elevation) * torch.sin(azimuth),
            distance * torch.sin(elevation),
            -distance * torch.cos(elevation) * torch.cos(azimuth)])
