"""Wavefront .obj writer -- reference neural_renderer/save_obj.py:10-191, same text format.

With `textures`, the [Nf,ts,ts,ts,3] texture cubes are exported as one atlas image (`<name>.png`, 16x16-pixel tile per
face) plus `<name>.mtl`; the atlas is rendered by the HIP kernels of `nr_create_texture_image` (csrc/nr_texture_io.hip, the
reference's two CUDA kernels of save_obj.py:32-146) and needs the GPU, like the reference's."""
import os

import numpy as np


def create_texture_image(textures, texture_size_out=16, device='cuda'):
    """-> (image [tile_h*tso, tile_w*tso, 3] float32 flipped vertically, uv triangles [Nf,3,2] in [0,1]) -- save_obj.py:10-147."""
    import torch

    from . import _lib

    textures = np.ascontiguousarray(textures, dtype=np.float32)
    num_faces, texture_size_in = textures.shape[:2]
    tile_width = int((num_faces - 1.) ** 0.5) + 1
    tile_height = int((num_faces - 1.) / tile_width) + 1
    height, width = tile_height * texture_size_out, tile_width * texture_size_out

    vertices = np.zeros((num_faces, 3, 2), 'float32')  # [:, :, XY] tile triangles in atlas pixels, :16-25
    face_nums = np.arange(num_faces)
    column = face_nums % tile_width
    row = face_nums // tile_width
    vertices[:, 0, 0] = column * texture_size_out
    vertices[:, 0, 1] = row * texture_size_out
    vertices[:, 1, 0] = column * texture_size_out
    vertices[:, 1, 1] = (row + 1) * texture_size_out - 1
    vertices[:, 2, 0] = (column + 1) * texture_size_out - 1
    vertices[:, 2, 1] = (row + 1) * texture_size_out - 1

    lib = _lib.load()
    dev = torch.device(device)
    with torch.cuda.device(dev):
        image_d = torch.empty((height, width, 3), dtype=torch.float32, device=dev)
        vertices_d = torch.from_numpy(vertices).to(dev)
        textures_d = torch.from_numpy(textures).to(dev)
        _lib.check(lib.nr_create_texture_image(textures_d.data_ptr(), vertices_d.data_ptr(), image_d.data_ptr(), num_faces,
                                               texture_size_in, texture_size_out, tile_width, tile_height,
                                               torch.cuda.current_stream(dev).cuda_stream), 'nr_create_texture_image')
        image = image_d.cpu().numpy()
    vertices[:, :, 0] /= (width - 1)   # :140-141
    vertices[:, :, 1] /= (height - 1)
    return image[::-1, ::1], vertices  # :143


def _toimage(image, cmin=0.0, cmax=1.0):
    """scipy.misc.toimage(image, cmin=0, cmax=1) (save_obj.py:158): (x - cmin) * 255 / (cmax - cmin), clip, + 0.5 -> uint8."""
    from PIL import Image
    data = (np.asarray(image, dtype=np.float64) - cmin) * (255.0 / (cmax - cmin))
    return Image.fromarray((data.clip(0, 255) + 0.5).astype(np.uint8))


def save_obj(filename, vertices, faces, textures=None):
    """Writes `filename` (and, with textures [Nf,ts,ts,ts,3], `<stem>.mtl` + `<stem>.png`) in the reference's text format
    (save_obj.py:150-191): a 3-line header, `v %.8f` lines, then either `f a b c` or `vt` lines + `usemtl` + `f a/ta b/tb c/tc`
    with three private uv entries per face."""
    assert vertices.ndim == 2
    assert faces.ndim == 2
    vertices = np.asarray(vertices)
    faces = np.asarray(faces)
    stem = filename[:-4]
    out = ['# %s\n' % os.path.basename(filename), '#\n', '\n']

    if textures is None:
        out += ['v %.8f %.8f %.8f\n' % (v[0], v[1], v[2]) for v in vertices]
        out.append('\n')
        out += ['f %d %d %d\n' % (a + 1, b + 1, c + 1) for a, b, c in faces]
    else:
        material = 'material_1'
        atlas, uv = create_texture_image(textures)
        _toimage(atlas, cmin=0, cmax=1).save(stem + '.png')
        with open(stem + '.mtl', 'w') as f:
            f.write('newmtl %s\nmap_Kd %s\n' % (material, os.path.basename(stem + '.png')))
        out.append('mtllib %s\n\n' % os.path.basename(stem + '.mtl'))
        out += ['v %.8f %.8f %.8f\n' % (v[0], v[1], v[2]) for v in vertices]
        out.append('\n')
        out += ['vt %.8f %.8f\n' % (t[0], t[1]) for t in uv.reshape((-1, 2))]
        out.append('\n')
        out.append('usemtl %s\n' % material)
        out += ['f %d/%d %d/%d %d/%d\n' % (a + 1, 3 * i + 1, b + 1, 3 * i + 2, c + 1, 3 * i + 3)
                for i, (a, b, c) in enumerate(faces)]
        out.append('\n')

    with open(filename, 'w') as f:
        f.writelines(out)
