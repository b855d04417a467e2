"""3-vector cross product per row -- reference neural_renderer/cross.py:58 (its CUDA kernel K9 is
off the hot path; stock torch reproduces it)."""
import torch


def cross(a, b):
    assert a.dim() == 2 and b.dim() == 2 and a.shape[1] == 3 and b.shape == a.shape
    return torch.cross(a, b, dim=1)
