"""Vertex gather: [bs, nv, 3] x [bs, nf, 3] -> [bs, nf, 3, 3] -- reference vertices_to_faces.py:4-21.

On the GPU both directions are HIP kernels of libnr_hip.so (`nr_vertices_to_faces`, `nr_vertices_to_faces_backward`):
the backward -- the face->vertex gradient scatter, Chainer's get_item backward in the reference -- uses hardware float
atomics instead of torch's sort-based `index_put_(accumulate=True)`, which alone cost more than the whole rasterizer
(scripts/glue_profile.py).  CPU tensors (host-side glue, tests) take the plain torch path."""
import torch

from . import _lib, _util


class _VerticesToFaces(torch.autograd.Function):
    @staticmethod
    def forward(ctx, vertices, faces):
        lib = _lib.load()
        v = vertices.detach().contiguous()
        f = faces.detach().to(torch.int32).contiguous()
        B, Nv = v.shape[:2]
        Nf = f.shape[1]
        out = torch.empty((B, Nf, 3, 3), dtype=torch.float32, device=v.device)
        with torch.cuda.device(v.device):
            _lib.check(lib.nr_vertices_to_faces(v.data_ptr(), f.data_ptr(), out.data_ptr(), B, Nv, Nf, 1,
                                                torch.cuda.current_stream(v.device).cuda_stream),
                       'nr_vertices_to_faces')
        ctx.save_for_backward(f)
        ctx.dims = (B, Nv, Nf)
        return out

    @staticmethod
    def backward(ctx, grad_faces):
        lib = _lib.load()
        f, = ctx.saved_tensors
        B, Nv, Nf = ctx.dims
        g = grad_faces.contiguous()
        grad_vertices = torch.empty((B, Nv, 3), dtype=torch.float32, device=g.device)
        with torch.cuda.device(g.device):
            _lib.check(lib.nr_vertices_to_faces_backward(g.data_ptr(), f.data_ptr(), grad_vertices.data_ptr(), B, Nv, Nf,
                                                         1, torch.cuda.current_stream(g.device).cuda_stream),
                       'nr_vertices_to_faces_backward')
        return grad_vertices, None


def vertices_to_faces(vertices, faces):
    """
    :param vertices: [batch size, number of vertices, 3]
    :param faces: [batch size, number of faces, 3)
    :return: [batch size, number of faces, 3, 3]
    """
    assert vertices.dim() == 3
    assert faces.dim() == 3
    assert vertices.shape[0] == faces.shape[0]
    assert vertices.shape[2] == 3
    assert faces.shape[2] == 3

    if vertices.is_cuda and vertices.dtype == torch.float32:
        _util.check_face_indices(faces, vertices.shape[1], vertices.device)  # IndexError / ValueError, cached per tensor
        return _VerticesToFaces.apply(vertices, faces)
    bs, nv = vertices.shape[:2]
    faces = faces.long() + (torch.arange(bs, device=vertices.device, dtype=torch.long) * nv)[:, None, None]
    vertices = vertices.reshape((bs * nv, 3))
    return vertices[faces]
