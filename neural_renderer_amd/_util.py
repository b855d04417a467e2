import numpy as np
import torch


def normalize(x, eps=1e-5):
    """chainer.functions.normalize: x / (||x||_2 + eps) along axis 1 (NOT torch.nn.functional.normalize,
    which divides by max(norm, eps)) -- used by reference look_at.py:30-32, look.py:29-31, lighting.py:40."""
    return x / (torch.sqrt(torch.sum(x * x, dim=1, keepdim=True)) + eps)


def as_tensor_like(value, ref, dtype=torch.float32):
    """list / tuple / ndarray / tensor -> tensor on ref's device (differentiable if it already is a tensor)."""
    if torch.is_tensor(value):
        return value.to(device=ref.device, dtype=dtype)
    return torch.as_tensor(np.asarray(value, dtype=np.float32), device=ref.device).to(dtype)
