"""`-m gpu`: pins for the two gradients the reference itself never tests (K7 backward_textures has no test, the
only K8 backward_depth_map test reads an empty batch slot) -- independent of the oracle's restatement:

  K7  rgb_map is LINEAR in the textures (rasterize.py:398-426 with a zero background), so grad_textures must be the exact
      adjoint: <g, rgb(T)> == <grad_textures(g), T> for any T, g (to float rounding), and a single hot texel must receive
      the sum of its sampling weights times g.
  K8  (a) the intent of reference tests/test_rasterize_depth.py:60-93 with the indexing fixed (the data sits in batch slot 2),
      through `Renderer.render_depth` on the GPU against finite differences of the rendered depth; (b) the gradient of a
      weighted sum of the depths of a face's interior pixels against central differences of the forward's own formula
      (:317-330) evaluated in float64.
All through the torch-facing API, i.e. through libnr_hip.so."""
import numpy as np
import pytest
import torch

import helpers as H

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('ts', [2, 4, 7])
def test_k7_grad_textures_is_the_exact_adjoint_of_the_sampling(ts):
    import neural_renderer_amd as nr
    faces, _ = H.teapot_views(3, 96)
    rng = np.random.default_rng(70 + ts)
    B, F = faces.shape[:2]
    S = 96
    ft = torch.tensor(faces, device='cuda')

    def render(tex):
        fn = nr.Rasterize(S, 0.1, 100, 1e-3, (0, 0, 0), True, False, False)
        fn.fix_batch_z = True
        return fn(ft, tex)[0]

    T = torch.tensor(rng.uniform(-1, 1, (B, F, ts, ts, ts, 3)).astype(np.float32), device='cuda', requires_grad=True)
    g = torch.tensor(rng.normal(size=(B, S, S, 3)).astype(np.float32), device='cuda')
    rgb = render(T)
    rgb.backward(g)
    lhs = float((g.double() * rgb.detach().double()).sum())
    rhs = float((T.grad.double() * T.detach().double()).sum())
    scale = float((g.double().abs() * rgb.detach().double().abs()).sum())
    assert abs(lhs - rhs) <= 2e-6 * scale, (lhs, rhs, scale)
    # a second, independent direction: the adjoint identity must hold for textures the gradient was not computed at
    T2 = torch.tensor(rng.uniform(-1, 1, T.shape).astype(np.float32), device='cuda')
    lhs2 = float((g.double() * render(T2).double()).sum())
    rhs2 = float((T.grad.double() * T2.double()).sum())
    assert abs(lhs2 - rhs2) <= 2e-6 * scale, (lhs2, rhs2)
    # single hot texels: rgb(e) is the image of that texel's sampling weights, so grad[texel] = sum_pixels weight * g
    fi = None
    for _ in range(6):
        b, f = int(rng.integers(B)), int(rng.integers(F))
        idx = tuple(int(x) for x in rng.integers(0, ts, 3)) + (int(rng.integers(3)),)
        E = torch.zeros_like(T2)
        E[(b, f) + idx] = 1.0
        w_img = render(E)
        expect = float((w_img.double() * g.double()).sum())
        got = float(T.grad[(b, f) + idx])
        assert abs(got - expect) <= 1e-5 * max(1.0, abs(expect)), (b, f, idx, got, expect)


def test_k8_reference_depth_test_with_the_indexing_fixed():
    """reference tests/test_rasterize_depth.py:60-93: one triangle, orthographic, camera_mode 'none', 64x64, loss on one
    pixel, forward differences with step 1e-3, atol 1e-3 -- reading batch slot 2, where tests/utils.py puts the data."""
    import neural_renderer_amd as nr
    vertices = np.array([[-0.9, -0.9, 2.], [-0.8, 0.8, 1.], [0.8, 0.8, 0.5]], np.float32)
    faces = np.array([[0, 1, 2]], np.int32)
    vb, fb = H.to_minibatch((vertices, faces))
    renderer = nr.Renderer()
    renderer.image_size = 64
    renderer.anti_aliasing = False
    renderer.perspective = False
    renderer.camera_mode = 'none'
    fbt = torch.tensor(fb, device='cuda')

    def loss_of(v):
        images = renderer.render_depth(v, fbt)
        return torch.sum(torch.square(images[2, 15, 20] - 1)), images

    vt = torch.tensor(vb, device='cuda', requires_grad=True)
    loss, images = loss_of(vt)
    assert float(images[2, 15, 20]) < 100  # the probe pixel lies on the face
    loss.backward()
    grad = vt.grad[2].cpu().numpy()
    assert np.all(vt.grad[[0, 1, 3]].cpu().numpy() == 0)
    grad2 = np.zeros_like(grad)
    h = 1e-3
    for i in range(3):
        for j in range(3):
            vp, vm = vb.copy(), vb.copy()
            vp[2, i, j] += h
            vm[2, i, j] -= h
            lp = float(loss_of(torch.tensor(vp, device='cuda'))[0])
            lm = float(loss_of(torch.tensor(vm, device='cuda'))[0])
            grad2[i, j] = (lp - lm) / (2 * h)
    assert np.abs(grad).max() > 1e-2
    np.testing.assert_allclose(grad, grad2, atol=1e-3)


def _depth_f64(face, S, pixels):
    """The forward's depth formula (rasterize.py:258-269, :317-330) for the given (xi, yi) pixels, in float64."""
    p = 0.5 * (face[:, :2] * S + S - 1)
    m = np.array([[p[0, 0], p[1, 0], p[2, 0]], [p[0, 1], p[1, 1], p[2, 1]], [1.0, 1.0, 1.0]])
    inv = np.linalg.inv(m)
    pts = np.stack((pixels[:, 0], pixels[:, 1], np.ones(len(pixels))), axis=0)
    w = inv @ pts
    w = np.clip(w, 0, 1)
    w = w / w.sum(0, keepdims=True)
    return 1.0 / (w[0] / face[0, 2] + w[1] / face[1, 2] + w[2] / face[2, 2])


def test_k8_against_float64_central_differences_of_the_depth_formula():
    import neural_renderer_amd as nr
    rng = np.random.default_rng(88)
    S = 96
    # a few large, well-separated front faces with distinct depths per vertex
    faces = np.zeros((2, 3, 3, 3), np.float32)
    for b in range(2):
        for f in range(3):
            c = np.array([-0.55 + 0.55 * f, -0.4 + 0.5 * b])
            tri = np.array([[-0.22, -0.2], [0.24, -0.17], [0.03, 0.26]]) + rng.normal(size=(3, 2)) * 0.02 + c
            faces[b, f, :, :2] = tri
            faces[b, f, :, 2] = rng.uniform(0.8, 3.0, 3)
    ft = torch.tensor(faces, device='cuda', requires_grad=True)
    fn = nr.Rasterize(S, 0.1, 100, 1e-3, None, False, False, True)
    depth = fn(ft)[2]
    fi = fn.face_index_map.cpu().numpy()
    assert (fi >= 0).sum() > 500
    # weights only on pixels whose 3x3 neighbourhood belongs to the same face (coverage cannot change under a tiny move)
    coef = np.zeros((2, S, S), np.float64)
    for b in range(2):
        for f in range(3):
            own = fi[b] == f
            inner = own.copy()
            for dy in (-1, 0, 1):
                for dx in (-1, 0, 1):
                    inner &= np.roll(np.roll(own, dy, 0), dx, 1)
            assert inner.sum() > 20
            coef[b][inner] = rng.normal(size=int(inner.sum()))
    depth.backward(torch.tensor(coef.astype(np.float32), device='cuda'))
    grad = ft.grad.cpu().numpy().astype(np.float64)
    ref = np.zeros_like(grad)
    h = 1e-6
    for b in range(2):
        for f in range(3):
            ys, xs = np.nonzero((fi[b] == f) & (coef[b] != 0))
            pix = np.stack((xs, ys), axis=1).astype(np.float64)
            cw = coef[b][ys, xs].astype(np.float32).astype(np.float64)
            base = faces[b, f].astype(np.float64)
            for k in range(3):
                for d in range(3):
                    fp, fm = base.copy(), base.copy()
                    fp[k, d] += h
                    fm[k, d] -= h
                    ref[b, f, k, d] = ((cw * _depth_f64(fp, S, pix)).sum() - (cw * _depth_f64(fm, S, pix)).sum()) / (2 * h)
    assert np.abs(ref).max() > 1.0
    assert H.rel_err(grad, ref) <= 1e-4, H.rel_err(grad, ref)
