#!/usr/bin/env python3
"""Materialise examples/data/ (teapot.obj and the reference images of examples 2-4) from the golden fixture
archive tests/golden/reference_fixtures.npz (made once from the reference's own data files by
tests/golden/make_golden.py).  The .obj is written by this package's save_obj with full float32 precision."""
import os
import sys

import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import neural_renderer_amd as nr  # noqa: E402


def main():
    g = np.load(os.path.join(os.path.dirname(HERE), 'tests', 'golden', 'reference_fixtures.npz'))
    out = os.path.join(HERE, 'data')
    os.makedirs(out, exist_ok=True)
    nr.save_obj(os.path.join(out, 'teapot.obj'), g['teapot_vertices_raw'], g['teapot_faces'])
    for name in ('example2_ref', 'example3_ref', 'example4_ref'):
        Image.fromarray(g[name]).save(os.path.join(out, name + '.png'))
    print('wrote', sorted(os.listdir(out)))


if __name__ == '__main__':
    main()
