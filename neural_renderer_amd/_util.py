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


_INDEX_ATTR = '_nr_index_ok'  # verdict stashed on the tensor OBJECT: it dies with the tensor (a cache keyed on data_ptr would
                               # be hit by an unrelated tensor that the caching allocator places at the same address)


def check_face_indices(faces, num_vertices, device=None):
    """Vertex indices must lie in [0, num_vertices) and live on the vertices' device (the reference's get_item raises on
    a bad index; a raw kernel would read / atomically add outside the buffers).  The verdict is remembered on the index
    tensor itself as (version counter, num_vertices), so an optimisation loop that passes the same tensor pays the device
    round trip once, and an in-place edit or another tensor is checked again."""
    if device is not None and faces.device != device:
        raise ValueError('faces are on %s but vertices on %s' % (faces.device, device))
    stamp = (faces._version, int(num_vertices), tuple(faces.shape))
    if getattr(faces, _INDEX_ATTR, None) == stamp:
        return
    if faces.numel():
        lo, hi = (int(x) for x in torch.aminmax(faces.detach()))
        if lo < 0 or hi >= num_vertices:
            raise IndexError('face vertex index out of range: [%d, %d] with %d vertices' % (lo, hi, num_vertices))
    try:
        setattr(faces, _INDEX_ATTR, stamp)
    except Exception:  # (a tensor subclass without __dict__: just check every time)
        pass
