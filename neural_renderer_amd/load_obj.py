"""Wavefront .obj loader -- reference neural_renderer/load_obj.py:9-197.

`load_obj(filename, normalization, texture_size, load_texture)` returns host arrays like the reference:
(vertices [Nv,3] float32, faces [Nf,3] int32[, textures [Nf,ts,ts,ts,3] float32]).  Texture baking (`load_texture=True`,
the reference's CUDA kernel of load_obj.py:87-144) runs as the HIP kernel `nr_load_textures` (csrc/nr_texture_io.hip) and
therefore needs the GPU, exactly like the reference's; there is no host fallback."""
import os

import numpy as np


def load_mtl(filename_mtl):
    """Diffuse colours (Kd) and texture file names (map_Kd) by material -- load_obj.py:9-22."""
    texture_filenames = {}
    colors = {}
    material_name = ''
    with open(filename_mtl) as f:
        for line in f:
            t = line.split()
            if len(t) == 0:
                continue
            if t[0] == 'newmtl':
                material_name = t[1]
            elif t[0] == 'map_Kd':
                texture_filenames[material_name] = t[1]
            elif t[0] == 'Kd':
                colors[material_name] = np.array([float(c) for c in t[1:4]], dtype=np.float32)
    return colors, texture_filenames


def _read_image(path):
    """[H,W,3] float32 in [0,1] (the reference: skimage.io.imread(...) / 255, load_obj.py:83)."""
    from PIL import Image
    return np.asarray(Image.open(path).convert('RGB'), dtype=np.float32) / np.float32(255.)


def load_textures(filename_obj, filename_mtl, texture_size, device='cuda'):
    """Texture cubes [Nf,ts,ts,ts,3] from the mesh's uv coordinates, materials and texture images -- load_obj.py:25-144."""
    import torch

    from . import _lib

    # uv coordinates (`vt`), uv indices of the fan-triangulated faces and the material in force at each face (:26-62)
    uv = []
    corner_ids = []
    material_names = []
    material_name = ''
    with open(filename_obj) as f:
        lines = f.readlines()
    for line in lines:
        t = line.split()
        if len(t) != 0 and t[0] == 'vt':
            uv.append([float(c) for c in t[1:3]])
    for line in lines:
        t = line.split()
        if len(t) == 0:
            continue
        if t[0] == 'f':
            ids = [int(c.split('/')[1]) if '/' in c else 0 for c in t[1:]]
            for i in range(len(ids) - 2):
                corner_ids.append((ids[0], ids[i + 1], ids[i + 2]))
                material_names.append(material_name)
        elif t[0] == 'usemtl':
            material_name = t[1]
    uv = np.vstack(uv).astype('float32')
    faces_uv = uv[np.vstack(corner_ids).astype('int32') - 1]      # a missing index (0 - 1) wraps to the last `vt`, as NumPy does
    wrap = 1 < faces_uv
    faces_uv[wrap] = faces_uv[wrap] % 1                            # :64

    colors, texture_filenames = load_mtl(filename_mtl)
    num_faces = faces_uv.shape[0]
    material_names = np.array(material_names)
    textures = np.zeros((num_faces, texture_size, texture_size, texture_size, 3), 'float32') + 0.5   # :69
    for name, color in colors.items():                             # :73-77
        textures[material_names == name] = color[None, None, None, None, :]
    if len(texture_filenames) == 0:
        return textures

    lib = _lib.load()
    dev = torch.device(device)
    with torch.cuda.device(dev):
        textures_d = torch.from_numpy(textures).to(dev)
        faces_uv_d = torch.from_numpy(np.ascontiguousarray(faces_uv)).to(dev)
        stream = torch.cuda.current_stream(dev).cuda_stream
        for name, filename_texture in texture_filenames.items():   # :80-143
            image = _read_image(os.path.join(os.path.dirname(filename_obj), filename_texture))
            image_d = torch.from_numpy(np.ascontiguousarray(image[::-1])).to(dev)    # vertical flip, :85
            is_update = torch.from_numpy((material_names == name).astype('int32')).to(dev)
            _lib.check(lib.nr_load_textures(image_d.data_ptr(), faces_uv_d.data_ptr(), is_update.data_ptr(),
                                            textures_d.data_ptr(), num_faces, texture_size, image.shape[0], image.shape[1],
                                            stream), 'nr_load_textures')
        return textures_d.cpu().numpy()                            # :144 (`textures.get()`)


def load_obj(filename_obj, normalization=True, texture_size=4, load_texture=False):
    """Returns (vertices [Nv,3] float32, faces [Nf,3] int32) and, with `load_texture`, textures [Nf,ts,ts,ts,3].
    Polygons are fan-triangulated (:167-175); with `normalization` the mesh is scaled into the unit cube centred at
    zero (:188-192)."""
    vertices = []
    faces = []
    with open(filename_obj) as f:
        lines = f.readlines()
    for line in lines:
        t = line.split()
        if len(t) == 0:
            continue
        if t[0] == 'v':
            vertices.append([float(v) for v in t[1:4]])
        elif t[0] == 'f':
            vs = t[1:]
            v0 = int(vs[0].split('/')[0])
            for i in range(len(vs) - 2):
                v1 = int(vs[i + 1].split('/')[0])
                v2 = int(vs[i + 2].split('/')[0])
                faces.append((v0, v1, v2))
    vertices = np.vstack(vertices).astype('float32')
    faces = np.vstack(faces).astype('int32') - 1

    textures = None
    if load_texture:  # :177-185
        for line in lines:
            if line.startswith('mtllib'):
                filename_mtl = os.path.join(os.path.dirname(filename_obj), line.split()[1])
                textures = load_textures(filename_obj, filename_mtl, texture_size)
        if textures is None:
            raise Exception('Failed to load textures.')

    if normalization:
        vertices -= vertices.min(0)[None, :]
        vertices /= np.abs(vertices).max()
        vertices *= 2
        vertices -= vertices.max(0)[None, :] / 2
    if load_texture:
        return vertices, faces, textures
    return vertices, faces
