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


_INDEX_OK = {}


def check_face_indices(faces, num_vertices, device=None):
    """Vertex indices must lie in [0, num_vertices) and live on the vertices' device (the reference's get_item raises on
    a bad index; a raw kernel would read / atomically add outside the buffers).  The verdict for a given index tensor is
    cached on (storage, shape, version), so an optimisation loop pays the device round trip once."""
    if device is not None and faces.device != device:
        raise ValueError('faces are on %s but vertices on %s' % (faces.device, device))
    key = (faces.data_ptr(), tuple(faces.shape), faces.dtype, faces._version, int(num_vertices), str(faces.device))
    if _INDEX_OK.get(key):
        return
    if faces.numel():
        lo, hi = (int(x) for x in torch.aminmax(faces.detach()))
        if lo < 0 or hi >= num_vertices:
            raise IndexError('face vertex index out of range: [%d, %d] with %d vertices' % (lo, hi, num_vertices))
    if len(_INDEX_OK) > 256:
        _INDEX_OK.clear()
    _INDEX_OK[key] = True
