"""Mesh holder with learnable vertices / textures -- reference neural_renderer/mesh.py:7-38 as an nn.Module."""
import torch
import torch.nn as nn

from .load_obj import load_obj


class Mesh(nn.Module):
    def __init__(self, filename_obj, texture_size=4, normalization=True):
        super(Mesh, self).__init__()
        vertices, faces = load_obj(filename_obj, normalization)
        self.vertices = nn.Parameter(torch.from_numpy(vertices))
        self.register_buffer('faces', torch.from_numpy(faces))
        self.num_vertices = self.vertices.shape[0]
        self.num_faces = self.faces.shape[0]
        shape = (self.num_faces, texture_size, texture_size, texture_size, 3)
        # mesh.py:22-24: chainer.initializers.Normal() = N(0, 0.05^2)
        self.textures = nn.Parameter(torch.randn(shape, dtype=torch.float32) * 0.05)
        self.texture_size = texture_size

    def get_batch(self, batch_size):
        # broadcast for minibatch (mesh.py:29-34)
        vertices = self.vertices[None].expand(batch_size, *self.vertices.shape)
        faces = self.faces[None].expand(batch_size, *self.faces.shape)
        textures = torch.sigmoid(self.textures[None].expand(batch_size, *self.textures.shape))
        return vertices, faces, textures

    def set_lr(self, lr_vertices, lr_textures):
        """Per-parameter learning-rate multipliers read by neural_renderer_amd.Adam (mesh.py:36-38)."""
        self.vertices.lr = lr_vertices
        self.textures.lr = lr_textures
