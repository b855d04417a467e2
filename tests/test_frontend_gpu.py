"""GPU parity of the two callers either side of the rasterizer:

  * nr_frontend_forward / _backward  (fill_back + lighting + look_at / look + perspective + vertices_to_faces,
    reference renderer.py:35-107) against (i) the oracle's NumPy restatement of that glue and (ii) the module-by-module
    torch path of neural_renderer_amd.Renderer, whose autograd gives the reference gradients;
  * nr_image_epilogue / _backward    (transpose + flip + 2x2 mean, reference rasterize.py:953-969) against the oracle.

Float tolerances: forward 1e-5 relative (the 3x3 rotation is applied in a fixed order, BLAS in the references), gradients
1e-4 relative to the largest component (float atomics / reductions in a different order).
"""
import numpy as np
import pytest
import torch

import helpers as H
from oracle import oracle as O

pytestmark = pytest.mark.gpu

RTOL = 1e-4


def _renderer(mode='look_at', perspective=True, fill_back=True, eye=None):
    import neural_renderer_amd as nr
    r = nr.Renderer()
    r.camera_mode = mode
    r.perspective = perspective
    r.fill_back = fill_back
    r.light_direction = [0.3, 0.8, -0.5]
    r.light_color_ambient = [1.0, 0.9, 0.8]
    r.light_color_directional = [0.7, 1.0, 0.6]
    r.light_intensity_ambient = 0.4
    r.light_intensity_directional = 0.6
    if mode == 'look':
        r.camera_direction = [0.2, -0.1, 1.0]
    if eye is not None:
        r.eye = eye
    return r


def _scene(B, ts, seed):
    rng = np.random.default_rng(seed)
    v, f = H.teapot()
    vb = (v[None] + rng.normal(scale=0.01, size=(B,) + v.shape)).astype(np.float32)
    fb = np.repeat(f[None], B, axis=0)
    tex = rng.uniform(0, 1, (B, f.shape[0], ts, ts, ts, 3)).astype(np.float32)
    eyes = np.array([O.get_points_from_angles(2.732, 20.0 + 5 * i, 70.0 * i) for i in range(B)], np.float32)
    return vb, fb, tex, eyes


@pytest.mark.parametrize('mode,perspective,fill_back,per_batch_eye,ts', [
    ('look_at', True, True, True, 2), ('look_at', True, False, False, 3), ('look', True, True, True, 2),
    ('look_at', False, True, False, 4), ('look', False, False, True, 2)])
def test_frontend_matches_torch_chain_and_oracle(mode, perspective, fill_back, per_batch_eye, ts):
    from neural_renderer_amd import frontend
    B = 3
    vb, fb, tex, eyes = _scene(B, ts, seed=71)
    eye_np = eyes if per_batch_eye else eyes[1]
    rng = np.random.default_rng(72)

    def run(fused):
        v = torch.tensor(vb, device='cuda', requires_grad=True)
        t = torch.tensor(tex, device='cuda', requires_grad=True)
        e = torch.tensor(eye_np, device='cuda', requires_grad=True)
        r = _renderer(mode, perspective, fill_back, e)
        f = torch.tensor(fb, device='cuda')
        assert frontend.fusable(r, v, f, t)
        faces, lit = frontend.project_and_light(r, v, f, t) if fused else r._frontend_torch(v, f, t)
        return v, t, e, faces, lit

    v1, t1, e1, faces1, lit1 = run(True)
    v0, t0, e0, faces0, lit0 = run(False)
    assert faces1.shape == faces0.shape and lit1.shape == lit0.shape
    np.testing.assert_allclose(faces1.detach().cpu().numpy(), faces0.detach().cpu().numpy(), rtol=1e-5, atol=2e-6)
    np.testing.assert_allclose(lit1.detach().cpu().numpy(), lit0.detach().cpu().numpy(), rtol=1e-6, atol=1e-7)

    # oracle restatement of the same glue (NumPy float32)
    f_all = np.concatenate((fb, fb[:, :, ::-1]), axis=1) if fill_back else fb
    t_all = np.concatenate((tex, tex.transpose((0, 1, 4, 3, 2, 5))), axis=1) if fill_back else tex
    lit_ref = O.lighting(O.vertices_to_faces(vb, f_all), t_all, 0.4, 0.6, [1.0, 0.9, 0.8], [0.7, 1.0, 0.6], [0.3, 0.8, -0.5])
    np.testing.assert_allclose(lit1.detach().cpu().numpy(), lit_ref, rtol=1e-6, atol=1e-7)
    if mode == 'look_at':
        vv = O.look_at(vb, eye_np)
        if perspective:
            vv = O.perspective(vv, 30)
        np.testing.assert_allclose(faces1.detach().cpu().numpy(), O.vertices_to_faces(vv, f_all), rtol=1e-5, atol=2e-6)

    # gradients: random cotangents through both paths
    gf = torch.tensor(rng.normal(size=tuple(faces0.shape)).astype(np.float32), device='cuda')
    gl = torch.tensor(rng.normal(size=tuple(lit0.shape)).astype(np.float32), device='cuda')
    torch.autograd.backward([faces1, lit1], [gf, gl])
    torch.autograd.backward([faces0, lit0], [gf, gl])
    assert H.rel_err(v1.grad.cpu().numpy(), v0.grad.cpu().numpy()) <= RTOL
    assert H.rel_err(t1.grad.cpu().numpy(), t0.grad.cpu().numpy()) <= RTOL
    assert e1.grad.shape == e0.grad.shape
    assert H.rel_err(e1.grad.cpu().numpy(), e0.grad.cpu().numpy()) <= RTOL


def test_frontend_without_textures_and_partial_gradients():
    """Silhouette / depth rendering: no lighting; and a call where only some inputs need gradients."""
    from neural_renderer_amd import frontend
    B = 2
    vb, fb, tex, eyes = _scene(B, 2, seed=73)
    f = torch.tensor(fb, device='cuda')
    g = None
    grads = []
    for fused in (True, False):
        v = torch.tensor(vb, device='cuda', requires_grad=True)
        r = _renderer(eye=eyes)
        faces, lit = frontend.project_and_light(r, v, f) if fused else r._frontend_torch(v, f)
        assert lit is None
        if g is None:
            g = torch.randn(faces.shape, device='cuda', generator=torch.Generator('cuda').manual_seed(1))
        faces.backward(g)
        grads.append(v.grad.cpu().numpy())
    assert H.rel_err(grads[0], grads[1]) <= RTOL

    # textures need a gradient, vertices do not; then the other way round with an unused texture output
    v = torch.tensor(vb, device='cuda')
    t = torch.tensor(tex, device='cuda', requires_grad=True)
    r = _renderer(eye=eyes[0])
    faces, lit = frontend.project_and_light(r, v, f, t)
    gl = torch.randn(lit.shape, device='cuda', generator=torch.Generator('cuda').manual_seed(2))
    lit.backward(gl)
    t0 = torch.tensor(tex, device='cuda', requires_grad=True)
    _, lit0 = r._frontend_torch(v, f, t0)
    lit0.backward(gl)
    assert H.rel_err(t.grad.cpu().numpy(), t0.grad.cpu().numpy()) <= RTOL
    v = torch.tensor(vb, device='cuda', requires_grad=True)
    faces, lit = frontend.project_and_light(r, v, f, torch.tensor(tex, device='cuda'))
    faces.sum().backward()   # the lit textures receive no gradient at all
    assert torch.isfinite(v.grad).all() and float(v.grad.abs().max()) > 0


def test_renderer_falls_back_to_torch_for_unfused_parameter_types():
    from neural_renderer_amd import frontend
    vb, fb, tex, eyes = _scene(2, 2, seed=74)
    v = torch.tensor(vb, device='cuda')
    f = torch.tensor(fb, device='cuda')
    t = torch.tensor(tex, device='cuda')
    r = _renderer(eye=eyes)
    assert frontend.fusable(r, v, f, t)
    r.light_direction = torch.tensor([[0., 1., 0.], [1., 0., 0.]], device='cuda')   # per-image light: torch path
    assert not frontend.fusable(r, v, f, t)
    r.image_size, r.anti_aliasing = 32, False
    assert r.render(v, f, t).shape == (2, 3, 32, 32)
    r = _renderer(eye=eyes)
    assert not frontend.fusable(r, v.cpu(), f.cpu(), t.cpu())
    r.camera_mode = 'none'
    assert not frontend.fusable(r, v, f, None)


@pytest.mark.parametrize('aa,S', [(True, 64), (False, 33), (True, 2), (False, 1)])
def test_image_epilogue_bit_exact(aa, S):
    """nr_image_epilogue: transposition, flip and 2x2 mean bit-identical to the oracle's; backward exact."""
    from neural_renderer_amd.rasterize import _ImageEpilogue
    rng = np.random.default_rng(81)
    B = 3
    rgb = rng.normal(size=(B, S, S, 3)).astype(np.float32)
    alpha = rng.normal(size=(B, S, S)).astype(np.float32)
    depth = rng.normal(size=(B, S, S)).astype(np.float32)
    ref = {'rgb': rgb.transpose((0, 3, 1, 2))[:, :, ::-1, :], 'alpha': alpha[:, ::-1, :], 'depth': depth[:, ::-1, :]}
    if aa:
        ref = {k: O._avg_pool2(x) for k, x in ref.items()}
    tr, ta, td = [torch.tensor(x, device='cuda', requires_grad=True) for x in (rgb, alpha, depth)]
    out = _ImageEpilogue.apply(tr, ta, td, aa)
    for got, k in zip(out, ('rgb', 'alpha', 'depth')):
        np.testing.assert_array_equal(got.detach().cpu().numpy(), np.ascontiguousarray(ref[k]), err_msg=k)
    # partial request + backward: only rgb and depth, depth unused by the loss
    o_rgb, o_alpha, o_depth = _ImageEpilogue.apply(tr, None, td, aa)
    assert o_alpha is None
    g = rng.normal(size=tuple(o_rgb.shape)).astype(np.float32)
    o_rgb.backward(torch.tensor(g, device='cuda'))
    up = np.repeat(np.repeat(g, 2, axis=-2), 2, axis=-1) * np.float32(0.25) if aa else g
    np.testing.assert_array_equal(tr.grad.cpu().numpy(), np.ascontiguousarray(up[:, :, ::-1, :].transpose((0, 2, 3, 1))))
    assert td.grad is None


@pytest.mark.parametrize('seed', [1, 2])
def test_fuzz_frontend_against_torch_chain(seed):
    """Random cameras, lights and meshes: the fused front-end against the module-by-module torch path, outputs and gradients."""
    from neural_renderer_amd import frontend
    import neural_renderer_amd as nr
    rng = np.random.default_rng(seed)
    for it in range(12):
        B, Nv, Nf = int(rng.integers(1, 4)), int(rng.integers(3, 40)), int(rng.integers(1, 60))
        ts = int(rng.choice([2, 3, 4]))
        vb = rng.normal(scale=0.5, size=(B, Nv, 3)).astype(np.float32)
        # three distinct vertices per face: with a repeated vertex the normal is cross(v, v), exactly 0 here and in NumPy but
        # a rounding residual in torch.cross (fused multiply-add), which chainer-style normalisation (|n| + 1e-5) then turns
        # into a 1e-4 difference of the light colour -- of a face that has no area and is never drawn
        fb = np.stack([rng.permutation(Nv)[:3] for _ in range(B * Nf)]).reshape(B, Nf, 3).astype(np.int32)
        tex = rng.uniform(0, 1, (B, Nf, ts, ts, ts, 3)).astype(np.float32)
        per_batch_eye = bool(rng.integers(0, 2))
        eye_np = (rng.normal(size=(B, 3)) * 0.5 + np.array([0, 0, -2.5])).astype(np.float32)
        if not per_batch_eye:
            eye_np = eye_np[0]

        def make(requires=True):
            r = nr.Renderer()
            r.camera_mode = ['look_at', 'look'][int(rng.integers(0, 2))]
            r.perspective = bool(rng.integers(0, 2))
            r.fill_back = bool(rng.integers(0, 2))
            r.viewing_angle = float(rng.choice([10, 30, 45.5]))
            r.camera_direction = (rng.normal(size=3) + np.array([0, 0, 2.0])).tolist()
            r.light_direction = rng.normal(size=3).tolist()
            r.light_color_ambient = rng.uniform(0, 1, 3).tolist()
            r.light_color_directional = rng.uniform(0, 1, 3).tolist()
            r.light_intensity_ambient = float(rng.choice([0.0, 0.3, 1.0]))
            r.light_intensity_directional = float(rng.choice([0.0, 0.5]))
            return r
        state = rng.bit_generator.state
        results = []
        for fused in (True, False):
            rng.bit_generator.state = state          # identical renderer settings for both paths
            r = make()
            v = torch.tensor(vb, device='cuda', requires_grad=True)
            t = torch.tensor(tex, device='cuda', requires_grad=True)
            e = torch.tensor(eye_np, device='cuda', requires_grad=True)
            r.eye = e
            f = torch.tensor(fb, device='cuda')
            assert frontend.fusable(r, v, f, t)
            faces, lit = frontend.project_and_light(r, v, f, t) if fused else r._frontend_torch(v, f, t)
            gen = torch.Generator('cuda').manual_seed(100 + it)
            gf = torch.randn(faces.shape, device='cuda', generator=gen)
            gl = torch.randn(lit.shape, device='cuda', generator=gen)
            torch.autograd.backward([faces, lit], [gf, gl])
            results.append([x.detach().cpu().numpy() for x in (faces, lit, v.grad, t.grad, e.grad)])
        a, b = results
        ok = np.isfinite(b[0])
        np.testing.assert_allclose(a[0][ok], b[0][ok], rtol=2e-5, atol=5e-6)
        np.testing.assert_allclose(a[1], b[1], rtol=1e-6, atol=1e-7)
        for k in (2, 3, 4):
            fin = np.isfinite(b[k]) & np.isfinite(a[k])
            assert fin.mean() > 0.9
            assert H.rel_err(a[k][fin], b[k][fin]) <= 2e-4, (it, k, H.rel_err(a[k][fin], b[k][fin]))
