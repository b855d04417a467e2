"""
Example 4. Finding camera parameters.
(reference examples/example4.py re-hosted on PyTorch-ROCm: renderer.eye is a learnable tensor.)
"""
import argparse

import numpy as np
import torch
import torch.nn as nn
import tqdm

import neural_renderer
from example_io import make_gif, read_image


class Model(nn.Module):
    def __init__(self, filename_obj, filename_ref=None):
        super(Model, self).__init__()
        vertices, faces = neural_renderer.load_obj(filename_obj)
        self.register_buffer('vertices', torch.from_numpy(vertices[None, :, :]))
        self.register_buffer('faces', torch.from_numpy(faces[None, :, :]))

        # create textures
        texture_size = 2
        self.register_buffer('textures', torch.ones((1, self.faces.shape[1], texture_size, texture_size, texture_size,
                                                     3), dtype=torch.float32))

        # load reference image
        if filename_ref is not None:
            ref = read_image(filename_ref)
            if ref.ndim == 3:
                ref = ref.max(-1)
            self.register_buffer('image_ref', torch.from_numpy((ref != 0).astype(np.float32)))
        else:
            self.image_ref = None

        # camera parameters
        self.camera_position = nn.Parameter(torch.tensor([6, 10, -14], dtype=torch.float32))

        # setup renderer
        self.renderer = neural_renderer.Renderer()

    def forward(self):
        self.renderer.eye = self.camera_position
        image = self.renderer.render_silhouettes(self.vertices, self.faces)
        loss = torch.sum(torch.square(image - self.image_ref[None, :, :]))
        return loss


def make_reference_image(filename_ref, filename_obj, device):
    from example_io import save_image
    model = Model(filename_obj).to(device)
    model.renderer.eye = neural_renderer.get_points_from_angles(2.732, 30, -15)
    with torch.no_grad():
        images = model.renderer.render(model.vertices, model.faces, model.textures)
    save_image(images.cpu().numpy()[0].transpose((1, 2, 0)), filename_ref)


def run():
    parser = argparse.ArgumentParser()
    parser.add_argument('-io', '--filename_obj', type=str, default='./examples/data/teapot.obj')
    parser.add_argument('-ir', '--filename_ref', type=str, default='./examples/data/example4_ref.png')
    parser.add_argument('-or', '--filename_output', type=str, default='./examples/data/example4_result.gif')
    parser.add_argument('-mr', '--make_reference_image', type=int, default=0)
    parser.add_argument('-g', '--gpu', type=int, default=0)
    parser.add_argument('--steps', type=int, default=1000)
    args = parser.parse_args()
    device = torch.device('cuda', args.gpu)

    if args.make_reference_image:
        make_reference_image(args.filename_ref, args.filename_obj, device)

    model = Model(args.filename_obj, args.filename_ref).to(device)
    optimizer = torch.optim.Adam(model.parameters(), lr=0.1)  # chainer Adam(alpha=0.1)
    frames = []
    # a one-image optimisation step is bound by the host: autograd's hand-over of the backward to its device thread is the
    # largest single item of it (0.55 -> 0.35 ms per step of example 2 on an MI355X box), so the backward stays on this thread
    # (a setting of this thread's autograd state: a context manager, so that nothing leaks into a caller of run())
    with neural_renderer.graph.backward_on_caller_thread():
        loop = tqdm.tqdm(range(args.steps))
        for i in loop:
            optimizer.zero_grad()
            loss = model()
            loss.backward()
            optimizer.step()
            with torch.no_grad():
                images = model.renderer.render(model.vertices, model.faces, model.textures)
            frames.append(images.cpu().numpy()[0].transpose((1, 2, 0)))
            loop.set_description('Optimizing (loss %.4f)' % float(loss))
            if float(loss) < 70:
                break
        print('stopped after %d steps, loss %.3f, camera %s' % (i + 1, float(loss), model.camera_position.tolist()))
        make_gif(frames, args.filename_output)


if __name__ == '__main__':
    run()
