"""
Example 1. Drawing a teapot from multiple viewpoints.
(reference examples/example1.py re-hosted on PyTorch-ROCm: same neural_renderer calls.)
"""
import argparse
import os

import numpy as np
import torch
import tqdm

import neural_renderer
from example_io import make_gif


def run():
    parser = argparse.ArgumentParser()
    parser.add_argument('-i', '--filename_input', type=str, default='./examples/data/teapot.obj')
    parser.add_argument('-o', '--filename_output', type=str, default='./examples/data/example1.gif')
    parser.add_argument('-g', '--gpu', type=int, default=0)
    parser.add_argument('--frames', type=int, default=90)
    args = parser.parse_args()
    device = torch.device('cuda', args.gpu)

    # other settings
    camera_distance = 2.732
    elevation = 30
    texture_size = 2

    # load .obj
    vertices, faces = neural_renderer.load_obj(args.filename_input)
    vertices = torch.from_numpy(vertices[None, :, :]).to(device)  # [num_vertices, XYZ] -> [batch_size=1, num_vertices, XYZ]
    faces = torch.from_numpy(faces[None, :, :]).to(device)  # [num_faces, 3] -> [batch_size=1, num_faces, 3]

    # create texture [batch_size=1, num_faces, texture_size, texture_size, texture_size, RGB]
    textures = torch.ones((1, faces.shape[1], texture_size, texture_size, texture_size, 3), dtype=torch.float32,
                          device=device)

    # create renderer
    renderer = neural_renderer.Renderer()

    # draw object
    frames = []
    for azimuth in tqdm.tqdm(np.linspace(0, 360, args.frames, endpoint=False)):
        renderer.eye = neural_renderer.get_points_from_angles(camera_distance, elevation, float(azimuth))
        images = renderer.render(vertices, faces, textures)  # [batch_size, RGB, image_size, image_size]
        image = images.detach().cpu().numpy()[0].transpose((1, 2, 0))  # [image_size, image_size, RGB]
        frames.append(image)
    make_gif(frames, args.filename_output)


if __name__ == '__main__':
    run()
