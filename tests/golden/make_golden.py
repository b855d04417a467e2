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


if __name__ == '__main__':
    main()
