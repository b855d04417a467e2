"""`-m gpu`: the four re-hosted example scripts drive the whole stack (load_obj -> Renderer -> HIP rasterizer ->
autograd -> torch.optim), a few optimisation steps each (BASELINE.json configs 2 and 3)."""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EX = os.path.join(ROOT, 'examples')


@pytest.fixture(scope='module')
def data_dir():
    sys.path.insert(0, EX)
    import make_data
    make_data.main()
    return os.path.join(EX, 'data')


def test_example1_turntable_batch16(data_dir):
    """BASELINE.json config 2: teapot, 16 azimuth views, 256x256 RGB + depth + silhouette forward + backward."""
    import neural_renderer as nr
    v, f = nr.load_obj(os.path.join(data_dir, 'teapot.obj'))
    B = 16
    vertices = torch.from_numpy(v)[None].repeat(B, 1, 1).cuda().requires_grad_(True)
    faces = torch.from_numpy(f)[None].repeat(B, 1, 1).cuda()
    textures = torch.ones((B, f.shape[0], 2, 2, 2, 3), device='cuda', requires_grad=True)
    renderer = nr.Renderer()
    eyes = [nr.get_points_from_angles(2.732, 30, 360.0 * i / B) for i in range(B)]
    renderer.eye = torch.tensor(eyes, dtype=torch.float32, device='cuda')
    rgb = renderer.render(vertices, faces, textures)
    sil = renderer.render_silhouettes(vertices, faces)
    depth = renderer.render_depth(vertices, faces)
    assert rgb.shape == (B, 3, 256, 256) and sil.shape == (B, 256, 256) and depth.shape == (B, 256, 256)
    cov = sil.detach().mean(dim=(1, 2)).cpu().numpy()
    assert (cov > 0.05).all() and (cov < 0.3).all()
    # anti-aliased silhouette is the 2x2 mean of a 0/1 map
    vals = np.unique(sil.detach().cpu().numpy())
    assert set(np.round(vals * 4).astype(int)) <= {0, 1, 2, 3, 4}
    (rgb.sum() + sil.sum() + (depth * (depth < 50)).sum()).backward()
    assert torch.isfinite(vertices.grad).all() and vertices.grad.abs().sum() > 0
    assert torch.isfinite(textures.grad).all() and textures.grad.abs().sum() > 0


def test_example2_vertex_optimisation(data_dir):
    import example2
    model = example2.Model(os.path.join(data_dir, 'teapot.obj'), os.path.join(data_dir, 'example2_ref.png')).cuda()
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    losses = []
    for _ in range(60):
        opt.zero_grad()
        loss = model()
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
    assert np.isfinite(losses).all()
    assert losses[-1] < 0.8 * losses[0], (losses[0], losses[-1])  # silhouette moves towards the rectangle


def test_example2_first_steps_match_the_oracle(data_dir):
    """BASELINE.json config 3 (example2: teapot -> rectangle silhouette loss, 256x256 with anti-aliasing, Adam): the loss
    curve and the vertices of the first optimisation steps against the same loop driven by the CPU oracle (its renderer,
    its backward through the epilogue / rasterizer / projection, a NumPy Adam).  Tolerances: loss 1e-5 relative,
    vertices 2e-6 absolute after 4 steps of size 1e-3."""
    import example2
    sys.path.insert(0, ROOT)
    from oracle import oracle as O
    model = example2.Model(os.path.join(data_dir, 'teapot.obj'), os.path.join(data_dir, 'example2_ref.png')).cuda()
    v = model.vertices.detach().cpu().numpy().copy()
    f = model.faces.cpu().numpy()
    ref = model.image_ref.cpu().numpy()
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    steps, losses = 4, []
    for _ in range(steps):
        opt.zero_grad()
        loss = model()
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))

    eye = O.get_points_from_angles(2.732, 0, 90)
    renderer = O.Renderer()
    renderer.eye = eye
    m, s2 = np.zeros_like(v, dtype=np.float64), np.zeros_like(v, dtype=np.float64)
    ref_losses = []
    for t in range(1, steps + 1):
        faces = renderer.project(v, f)
        out = O.rasterize_rgbad(faces, None, 256, True, return_rgb=False, return_alpha=True, return_depth=False,
                                return_function=True)
        image = out['alpha']
        ref_losses.append(float(np.sum(np.square(image.astype(np.float64) - ref[None]))))
        g_faces, = O.rgbad_backward(out['function'], True, None, (2 * (image - ref[None])).astype(np.float32), None)
        g = O.project_backward(v, f, eye, g_faces)
        m = 0.9 * m + 0.1 * g                                    # torch.optim.Adam defaults
        s2 = 0.999 * s2 + 0.001 * g * g
        v = (v - 1e-3 * (m / (1 - 0.9 ** t)) / (np.sqrt(s2 / (1 - 0.999 ** t)) + 1e-8)).astype(np.float32)
    np.testing.assert_allclose(losses, ref_losses, rtol=1e-5)
    assert losses[-1] < losses[0]
    np.testing.assert_allclose(model.vertices.detach().cpu().numpy(), v, atol=2e-6)


def test_example3_texture_optimisation(data_dir):
    import example3
    np.random.seed(0)
    model = example3.Model(os.path.join(data_dir, 'teapot.obj'), os.path.join(data_dir, 'example3_ref.png')).cuda()
    opt = torch.optim.Adam(model.parameters(), lr=0.1, betas=(0.5, 0.999))
    losses = []
    for _ in range(30):
        opt.zero_grad()
        loss = model()
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
    assert np.isfinite(losses).all() and min(losses[-5:]) < losses[0]


def test_example4_camera_optimisation(data_dir):
    import example4
    model = example4.Model(os.path.join(data_dir, 'teapot.obj'), os.path.join(data_dir, 'example4_ref.png')).cuda()
    opt = torch.optim.Adam(model.parameters(), lr=0.1)
    losses = []
    for _ in range(80):
        opt.zero_grad()
        loss = model()
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
    assert np.isfinite(losses).all()
    assert model.camera_position.grad is not None
    assert min(losses) < 0.7 * losses[0], (losses[0], min(losses))
