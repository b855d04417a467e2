"""
Example 3. Optimizing textures.
(reference examples/example3.py re-hosted on PyTorch-ROCm.)
"""
import argparse

import numpy as np
import torch
import torch.nn as nn
import tqdm

import neural_renderer
from example_io import make_gif, read_image


class Model(nn.Module):
    def __init__(self, filename_obj, filename_ref):
        super(Model, self).__init__()
        vertices, faces = neural_renderer.load_obj(filename_obj)
        self.register_buffer('vertices', torch.from_numpy(vertices[None, :, :]))
        self.register_buffer('faces', torch.from_numpy(faces[None, :, :]))

        # create textures
        texture_size = 4
        self.textures = nn.Parameter(torch.zeros((1, self.faces.shape[1], texture_size, texture_size, texture_size, 3),
                                                 dtype=torch.float32))

        # load reference image [RGB, H, W]
        ref = read_image(filename_ref)[:, :, :3].transpose((2, 0, 1))
        self.register_buffer('image_ref', torch.from_numpy(np.ascontiguousarray(ref)))

        # setup renderer
        renderer = neural_renderer.Renderer()
        renderer.perspective = False
        renderer.light_intensity_directional = 0.0
        renderer.light_intensity_ambient = 1.0
        self.renderer = renderer

    def forward(self):
        self.renderer.eye = neural_renderer.get_points_from_angles(2.732, 0, float(np.random.uniform(0, 360)))
        image = self.renderer.render(self.vertices, self.faces, torch.tanh(self.textures))
        loss = torch.sum(torch.square(image - self.image_ref[None]))
        return loss


def run():
    parser = argparse.ArgumentParser()
    parser.add_argument('-io', '--filename_obj', type=str, default='./examples/data/teapot.obj')
    parser.add_argument('-ir', '--filename_ref', type=str, default='./examples/data/example3_ref.png')
    parser.add_argument('-or', '--filename_output', type=str, default='./examples/data/example3_result.gif')
    parser.add_argument('-g', '--gpu', type=int, default=0)
    parser.add_argument('--steps', type=int, default=300)
    args = parser.parse_args()
    device = torch.device('cuda', args.gpu)

    model = Model(args.filename_obj, args.filename_ref).to(device)
    optimizer = torch.optim.Adam(model.parameters(), lr=0.1, betas=(0.5, 0.999))  # Adam(alpha=0.1, beta1=0.5)
    # a one-image optimisation step is bound by the host: autograd's hand-over of the backward to its device thread is the
    # largest single item of it (0.55 -> 0.35 ms per step of example 2 on an MI355X box), so the backward stays on this thread
    # (a setting of this thread's autograd state: a context manager, so that nothing leaks into a caller of run())
    with neural_renderer.graph.backward_on_caller_thread():
        loop = tqdm.tqdm(range(args.steps))
        for _ in loop:
            loop.set_description('Optimizing')
            optimizer.zero_grad()
            loss = model()
            loss.backward()
            optimizer.step()
        print('final loss %.3f' % float(loss))

    # draw object
    frames = []
    for azimuth in tqdm.tqdm(range(0, 360, 4), desc='Drawing'):
        model.renderer.eye = neural_renderer.get_points_from_angles(2.732, 0, azimuth)
        with torch.no_grad():
            images = model.renderer.render(model.vertices, model.faces, torch.tanh(model.textures))
        frames.append(images.cpu().numpy()[0].transpose((1, 2, 0)))
    make_gif(frames, args.filename_output)


if __name__ == '__main__':
    run()
