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


_INDEX_ATTR = '_nr_index_ok'  # verdict stashed on a tensor OBJECT: it dies with the tensor (a cache keyed on data_ptr would
                               # be hit by an unrelated tensor that the caching allocator places at the same address)


def _stamp(t, num_vertices):
    return (t._version, int(num_vertices), tuple(t.shape), tuple(t.stride()), t.storage_offset())


def check_face_indices(faces, num_vertices, device=None):
    """Vertex indices must lie in [0, num_vertices) and live on the vertices' device (the reference's get_item raises on
    a bad index; a raw kernel would read / atomically add outside the buffers).  The verdict is remembered as (version
    counter, num_vertices, geometry) on the index tensor itself -- and, for a VIEW (`faces[None]`, `.expand`, a slice built
    anew every step), on the tensor it is a view of, whose elements are a superset and whose object lives on: an
    optimisation loop pays the device round trip once, and an in-place edit (also through the view: views share the
    version counter) or another tensor is checked again.  Inside a HIP-graph capture nothing can be read back, so an
    index tensor without a verdict raises there: run one eager step first."""
    if device is not None and faces.device != device:
        raise ValueError('faces are on %s but vertices on %s' % (faces.device, device))
    stamp = _stamp(faces, num_vertices)
    if getattr(faces, _INDEX_ATTR, None) == stamp:
        return
    base = faces._base
    if base is not None and getattr(base, _INDEX_ATTR, None) == _stamp(base, num_vertices):
        return
    if faces.numel():
        if faces.is_cuda and torch.cuda.is_current_stream_capturing():
            raise RuntimeError('face indices cannot be range-checked while a HIP graph is being captured (the check reads two '
                               'values back): call the renderer once with this index tensor before the capture')
        for t in ((base, faces) if base is not None else (faces,)):
            lo, hi = (int(x) for x in torch.aminmax(t.detach()))
            if 0 <= lo and hi < num_vertices:
                try:
                    setattr(t, _INDEX_ATTR, _stamp(t, num_vertices))
                except Exception:  # (a tensor subclass without __dict__: just check every time)
                    pass
                return
        raise IndexError('face vertex index out of range: [%d, %d] with %d vertices' % (lo, hi, num_vertices))
