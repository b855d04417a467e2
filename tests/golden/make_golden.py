#!/usr/bin/env python3
"""Generate tests/golden/reference_fixtures.npz from the reference's own test data.

Run ONCE in the authoring container (where /root/reference is mounted read-only):

    python tests/golden/make_golden.py

The reference (hiroharu-kato/neural_renderer @ 1.1.3) is Python 2 + Chainer + CuPy + CUDA and
cannot be imported or executed here, so the golden vectors are the fixtures its own test-suite
ships, re-encoded losslessly into one .npz so that they travel to the GPU box:

  teapot_vertices_raw  float32 [1292,3]  `v` lines of tests/data/teapot.obj, *before* normalisation
                                         (reference loader: neural_renderer/load_obj.py:147-197)
  teapot_faces         int32   [2464,3]  `f` lines (0-based), fan-triangulated like load_obj.py:167-175
  teapot_blender       bool    [256,256] tests/data/teapot_blender.png -> `ref.min(-1) != 255`
                                         (tests/test_rasterize_silhouettes.py:29-33)
  test_depth           uint8   [256,256] tests/data/test_depth.png      (tests/test_rasterize_depth.py:54)
  test_rasterize1      uint8   [256,256,3] tests/data/test_rasterize1.png (tests/test_rasterize.py:32)
  test_rasterize2      uint8   [256,256,3] tests/data/test_rasterize2.png (tests/test_rasterize.py:50)
  example2_ref         uint8   [256,256]  examples/data/example2_ref.png  (examples/example2.py:34)
  example3_ref         uint8   [256,256,3] examples/data/example3_ref.png (examples/example3.py:27)
  example4_ref         uint8   [256,256]  examples/data/example4_ref.png  (examples/example4.py:31)
  silhouettes_case{1,2} bool   [64,64]    tests/data/rasterize_silhouettes_case{1,2}.png (unused by the
                                         reference tests; images of the two backward-test triangles)

and tests/golden/display_model.npz, the textured ShapeNet model of the reference's texture-loading test
(tests/test_load_obj.py:51-59: tests/data/4e49873292196f02574b5684eaec43e9/model.obj + model.mtl + images/*.jpg -> display.png):

  v            float32 [921,3]    `v` lines, before normalisation
  vt           float32 [Nt,2]     `vt` lines
  faces_v      int32   [3644,3]   vertex indices of the fan-triangulated faces (0-based)
  faces_vt     int32   [3644,3]   `vt` indices of the same corners (0-based)
  face_material int32  [3644]     index into `materials` of the `usemtl` in force at each face
  materials    str     [7]        material names in model.mtl order;  kd float64 [7,3] their `Kd`;  map_kd str [7] their
                                  `map_Kd` file ('' = none)
  image_<file> uint8   [H,W,3]    the texture JPEGs decoded HERE with PIL/libjpeg-turbo (the reference decoded them with its
                                  own skimage/libjpeg in 2018: decoders differ by a few levels, see tests/test_texture_io.py)
  display_png  uint8   [256,256,3] tests/data/display.png, the reference's render of that model (test_load_obj.py:59)

The hard-coded `grad_ref` constants of the backward tests are numbers in the reference's test
source (tests/test_rasterize_silhouettes.py:47-51,79-83); they are restated, with that citation, in
tests/test_oracle_golden.py rather than stored here.
"""
import os
import sys

import numpy as np
from PIL import Image

REF = os.environ.get('NR_REFERENCE', '/root/reference')
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'reference_fixtures.npz')


def read_obj_raw(path):
    vertices, faces = [], []
    with open(path) as f:
        for line in f:
            t = line.split()
            if not t:
                continue
            if t[0] == 'v':
                vertices.append([float(x) for x in t[1:4]])
            elif t[0] == 'f':
                idx = [int(s.split('/')[0]) for s in t[1:]]
                for i in range(len(idx) - 2):
                    faces.append((idx[0], idx[i + 1], idx[i + 2]))
    return np.asarray(vertices, np.float32), np.asarray(faces, np.int32) - 1


def img(path):
    return np.asarray(Image.open(os.path.join(REF, path)))


def main():
    if not os.path.isdir(REF):
        sys.exit('reference tree not found at %s' % REF)
    v, f = read_obj_raw(os.path.join(REF, 'tests/data/teapot.obj'))
    v2, f2 = read_obj_raw(os.path.join(REF, 'examples/data/teapot.obj'))
    assert np.array_equal(v, v2) and np.array_equal(f, f2), 'tests/ and examples/ teapots differ'
    blender = img('tests/data/teapot_blender.png')
    out = dict(
        teapot_vertices_raw=v,
        teapot_faces=f,
        teapot_blender=(blender[..., :3].min(-1) != 255) if blender.ndim == 3 else (blender != 255),
        test_depth=img('tests/data/test_depth.png'),
        test_rasterize1=img('tests/data/test_rasterize1.png'),
        test_rasterize2=img('tests/data/test_rasterize2.png'),
        example2_ref=img('examples/data/example2_ref.png'),
        example3_ref=img('examples/data/example3_ref.png'),
        example4_ref=img('examples/data/example4_ref.png'),
        silhouettes_case1=img('tests/data/rasterize_silhouettes_case1.png'),
        silhouettes_case2=img('tests/data/rasterize_silhouettes_case2.png'),
    )
    for k, a in out.items():
        print('%-22s %-8s %s' % (k, a.dtype, a.shape))
    np.savez_compressed(OUT, **out)
    print('wrote', OUT, os.path.getsize(OUT), 'bytes')


def display_model():
    root = os.path.join(REF, 'tests/data/4e49873292196f02574b5684eaec43e9')
    v, vt, fv, fvt, fmat = [], [], [], [], []
    names, kd, map_kd = [], {}, {}
    cur = ''
    for line in open(os.path.join(root, 'model.mtl')):
        t = line.split()
        if not t:
            continue
        if t[0] == 'newmtl':
            cur = t[1]
            names.append(cur)
        elif t[0] == 'Kd':
            kd[cur] = [float(x) for x in t[1:4]]
        elif t[0] == 'map_Kd':
            map_kd[cur] = t[1]
    cur = ''
    for line in open(os.path.join(root, 'model.obj')):
        t = line.split()
        if not t:
            continue
        if t[0] == 'v':
            v.append([float(x) for x in t[1:4]])
        elif t[0] == 'vt':
            vt.append([float(x) for x in t[1:3]])
        elif t[0] == 'usemtl':
            cur = t[1]
        elif t[0] == 'f':
            a = [int(c.split('/')[0]) for c in t[1:]]
            b = [int(c.split('/')[1]) if '/' in c else 0 for c in t[1:]]   # no uv: 0, i.e. -1 below (load_obj.py:44-56)
            for i in range(len(a) - 2):
                fv.append((a[0], a[i + 1], a[i + 2]))
                fvt.append((b[0], b[i + 1], b[i + 2]))
                fmat.append(names.index(cur))
    out = dict(
        v=np.asarray(v, np.float32), vt=np.asarray(vt, np.float32), faces_v=np.asarray(fv, np.int32) - 1,
        faces_vt=np.asarray(fvt, np.int32) - 1, face_material=np.asarray(fmat, np.int32), materials=np.asarray(names),
        kd=np.asarray([kd[n] for n in names], np.float64), map_kd=np.asarray([map_kd.get(n, '') for n in names]),
        display_png=img('tests/data/display.png'))
    for n in names:
        if n in map_kd:
            key = 'image_' + os.path.basename(map_kd[n])
            out[key] = np.asarray(Image.open(os.path.join(root, map_kd[n])).convert('RGB'))
    path = os.path.join(os.path.dirname(OUT), 'display_model.npz')
    for k, a in out.items():
        print('%-22s %-8s %s' % (k, a.dtype, a.shape))
    np.savez_compressed(path, **out)
    print('wrote', path, os.path.getsize(path), 'bytes')


if __name__ == '__main__':
    main()
    display_model()
