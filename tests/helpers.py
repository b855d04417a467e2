"""Shared test helpers: golden fixtures, teapot scene construction, comparison utilities."""
import math
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(HERE, 'golden', 'reference_fixtures.npz')

_golden = None


def golden():
    global _golden
    if _golden is None:
        _golden = dict(np.load(GOLDEN))
    return _golden


def teapot(normalization=True):
    """(vertices [1292,3] f32, faces [2464,3] i32) of the reference's teapot.obj (tests/test_load_obj.py:36)."""
    from oracle import oracle as O
    g = golden()
    v = g['teapot_vertices_raw']
    return (O.normalize_vertices(v) if normalization else v.copy()), g['teapot_faces'].copy()


def to_minibatch(data, batch_size=4, target_num=2):
    """Reference tests/utils.py:7-14: batch of zeros with the payload in slot `target_num`."""
    ret = []
    for d in data:
        d2 = np.repeat(np.expand_dims(np.zeros_like(d), 0), batch_size, axis=0)
        d2[target_num] = d
        ret.append(d2)
    return ret


def bytescale(x):
    """scipy.misc.imsave byte scaling used to write test_rasterize{1,2}.png."""
    x = np.asarray(x, np.float64)
    return np.floor(np.clip((x - x.min()) * 255.0 / (x.max() - x.min()), 0, 255) + 0.5).astype(np.uint8)


def teapot_views(batch, image_size=256, elevation=30.0, distance=2.732, fill_back=True):
    """The headline scene: teapot seen from `batch` azimuths 360*i/batch, elevation 30,
    distance 2.732 (examples/example1.py:26-27), through look_at + perspective(30) + vertices_to_faces.
    Returns faces [B, F, 3, 3] float32 in the rasterizer's input convention and world-space faces."""
    from oracle import oracle as O
    v, f = teapot()
    if fill_back:
        f = np.concatenate((f, f[:, ::-1]), axis=0)
    out = []
    for i in range(batch):
        eye = O.get_points_from_angles(distance, elevation, 360.0 * i / batch)
        vv = O.perspective(O.look_at(v[None], eye), 30.)
        out.append(O.vertices_to_faces(vv, f[None])[0])
    return np.stack(out).astype(np.float32), f


def random_scene(rng, batch, num_faces, spread=0.6, size=0.25, zmin=1.0, zmax=3.0):
    """Random triangle soup in the rasterizer's input convention: x,y in NDC, z = positive depth."""
    c = rng.uniform(-spread, spread, (batch, num_faces, 1, 3)).astype(np.float32)
    d = rng.uniform(-size, size, (batch, num_faces, 3, 3)).astype(np.float32)
    faces = c + d
    faces[..., 2] = rng.uniform(zmin, zmax, (batch, num_faces, 3)).astype(np.float32)
    return np.ascontiguousarray(faces, np.float32)


def rel_err(a, b, floor=None):
    """max |a-b| / max(|b|, floor); floor defaults to 1e-3 * max|b| (sum-order noise is relative to the
    magnitude of the partial sums, not of a result that may have cancelled)."""
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    if floor is None:
        floor = 1e-3 * (np.abs(b).max() if b.size else 1.0) + 1e-30
    return float((np.abs(a - b) / np.maximum(np.abs(b), floor)).max()) if b.size else 0.0


_display = None


def display_model():
    """tests/golden/display_model.npz: the textured ShapeNet model of the reference's tests/test_load_obj.py:51-59."""
    global _display
    if _display is None:
        _display = dict(np.load(os.path.join(os.path.dirname(GOLDEN), 'display_model.npz')))
    return _display


def write_display_model(dirpath):
    """Re-create model.obj / model.mtl / images/*.png from the fixture arrays (lossless: %.9g floats, PNG textures whose
    pixels are the JPEGs as decoded when the fixture was made).  Returns the .obj path."""
    from PIL import Image
    g = display_model()
    os.makedirs(os.path.join(dirpath, 'images'), exist_ok=True)
    with open(os.path.join(dirpath, 'model.mtl'), 'w') as f:
        for name, kd, tex in zip(g['materials'], g['kd'], g['map_kd']):
            f.write('newmtl %s\nKa 0.000000 0.000000 0.000000\nKd %.6f %.6f %.6f\n' % ((name,) + tuple(kd)))
            if tex:
                png = os.path.splitext(os.path.basename(str(tex)))[0] + '.png'
                Image.fromarray(g['image_' + os.path.basename(str(tex))]).save(os.path.join(dirpath, 'images', png))
                f.write('map_Kd ./images/%s\n' % png)
            f.write('\n')
    path = os.path.join(dirpath, 'model.obj')
    with open(path, 'w') as f:
        f.write('mtllib model.mtl\n')
        f.writelines('v %.9g %.9g %.9g\n' % tuple(v) for v in g['v'])
        f.writelines('vt %.9g %.9g\n' % tuple(t) for t in g['vt'])
        cur = -1
        for fv, ft, m in zip(g['faces_v'], g['faces_vt'], g['face_material']):
            if m != cur:
                cur = m
                f.write('usemtl %s\n' % g['materials'][m])
            if ft[0] < 0:
                f.write('f %d %d %d\n' % tuple(fv + 1))
            else:
                f.write('f %d/%d %d/%d %d/%d\n' % (fv[0] + 1, ft[0] + 1, fv[1] + 1, ft[1] + 1, fv[2] + 1, ft[2] + 1))
    return path
