#!/usr/bin/env python3
"""Materialise examples/data/ (teapot.obj and the reference images of examples 2-4) from the golden fixture
archive tests/golden/reference_fixtures.npz (made once from the reference's own data files by
tests/golden/make_golden.py).  The .obj is written with full float32 precision (`%.9g`; save_obj's `%.8f`, the
reference's format, would round the teapot's small coordinates)."""
import os

import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    g = np.load(os.path.join(os.path.dirname(HERE), 'tests', 'golden', 'reference_fixtures.npz'))
    out = os.path.join(HERE, 'data')
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, 'teapot.obj'), 'w') as f:
        f.writelines('v %.9g %.9g %.9g\n' % tuple(v) for v in g['teapot_vertices_raw'])
        f.writelines('f %d %d %d\n' % tuple(i + 1 for i in face) for face in g['teapot_faces'])
    for name in ('example2_ref', 'example3_ref', 'example4_ref'):
        Image.fromarray(g[name]).save(os.path.join(out, name + '.png'))
    print('wrote', sorted(os.listdir(out)))


if __name__ == '__main__':
    main()
