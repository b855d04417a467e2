"""Vertex gather: [bs, nv, 3] x [bs, nf, 3] -> [bs, nf, 3, 3] -- reference vertices_to_faces.py:4-21.
Its backward (index_add) is the face->vertex gradient scatter."""
import torch


def vertices_to_faces(vertices, faces):
    assert vertices.dim() == 3
    assert faces.dim() == 3
    assert vertices.shape[0] == faces.shape[0]
    assert vertices.shape[2] == 3
    assert faces.shape[2] == 3

    bs, nv = vertices.shape[:2]
    faces = faces.long() + (torch.arange(bs, device=vertices.device, dtype=torch.long) * nv)[:, None, None]
    vertices = vertices.reshape((bs * nv, 3))
    return vertices[faces]
