"""Per-face light colours instead of lit, duplicated textures (include/nr_hip.h: nr_face_light).

The reference's Renderer.render (renderer.py:77-103) hands the rasterizer textures that fill_back duplicated
(the copy with cube axes 0 and 2 exchanged, :79) and that lighting multiplied by one colour per face (lighting.py:50-51).
The `face_light` mode of the operator takes the original cubes and the colours.  These tests hold it against the
lit-texture path, which itself is parity-tested against the oracle (test_hip_parity.py, test_frontend_gpu.py):

  * every geometric output (face_index_map, alpha, depth) bit-identical -- textures do not enter the geometry;
  * rgb within LIGHT_ORDER = 1e-6 of the largest colour: the reference rounds light * texel per texel and sums the eight
    taps, the face_light mode rounds the sum and multiplies (measured 2e-7);
  * gradients (textures, light colours -> vertices, faces) within 1e-5 of their largest component, the tolerance of every
    reduction-order-dependent gradient in this suite (test_hip_parity.SAME_TERMS).
"""
import numpy as np
import pytest
import torch

import helpers as H
from oracle import oracle as O

pytestmark = pytest.mark.gpu

LIGHT_ORDER = 1e-6
GRAD_TOL = 1e-5


def _close(a, b, tol, what):
    a = a.detach().cpu().numpy().astype(np.float64)
    b = b.detach().cpu().numpy().astype(np.float64)
    assert a.shape == b.shape, what
    scale = max(np.abs(b).max(), 1e-30)
    err = np.abs(a - b).max() / scale
    assert err <= tol, '%s: %.3g > %.3g' % (what, err, tol)
    return err


def _faces_scene(B, Nf, seed, ground=False):
    """Random small triangles in NDC (+ optionally one screen-filling triangle per image: k_backward_big's case)."""
    rng = np.random.default_rng(seed)
    c = rng.uniform(-0.8, 0.8, (B, Nf, 1, 2))
    xy = c + rng.uniform(-0.25, 0.25, (B, Nf, 3, 2))
    z = rng.uniform(1.0, 3.0, (B, Nf, 3, 1))
    f = np.concatenate((xy, z), axis=3).astype(np.float32)
    if ground:
        f[:, 0] = np.array([[-0.95, -0.9, 4.0], [0.95, -0.85, 4.5], [0.0, 0.95, 5.0]], np.float32)
    return f


def _run(faces_np, tex_np, light_np, fill_back, lit_mode, S, anti_aliasing=False, upstream=None, rgbad=True):
    import neural_renderer_amd as nr
    f0 = torch.tensor(faces_np, device='cuda', requires_grad=True)
    t0 = torch.tensor(tex_np, device='cuda', requires_grad=True)
    l0 = torch.tensor(light_np, device='cuda', requires_grad=True)
    faces = torch.cat((f0, f0.flip(2)), dim=1) if fill_back else f0
    if lit_mode:
        out = nr.rasterize_rgbad(faces, t0, S, anti_aliasing, face_light=l0)
    else:
        t_all = torch.cat((t0, t0.permute(0, 1, 4, 3, 2, 5)), dim=1) if fill_back else t0
        out = nr.rasterize_rgbad(faces, t_all * l0[:, :, None, None, None, :], S, anti_aliasing)
    if upstream is None:
        return out
    loss = (out['rgb'] * upstream['rgb']).sum() + (out['alpha'] * upstream['alpha']).sum() \
        + (out['depth'] * upstream['depth']).sum()
    loss.backward()
    return out, f0.grad, t0.grad, l0.grad


@pytest.mark.parametrize('ts,fill_back,ground,S', [
    (2, True, False, 64), (2, True, True, 96), (2, False, False, 64), (4, True, False, 64), (4, True, True, 96),
    (3, False, True, 96), (6, True, False, 48), (9, True, False, 48), (13, False, False, 32), (14, True, False, 32)])
def test_face_light_operator_equals_lit_textures(ts, fill_back, ground, S):
    B, Nf = 3, 60
    rng = np.random.default_rng(100 + ts)
    faces = _faces_scene(B, Nf, 5 + ts, ground)
    tex = rng.uniform(0, 1, (B, Nf, ts, ts, ts, 3)).astype(np.float32)
    F = 2 * Nf if fill_back else Nf
    light = rng.uniform(0.2, 1.5, (B, F, 3)).astype(np.float32)
    up = {'rgb': torch.tensor(rng.normal(size=(B, 3, S, S)).astype(np.float32), device='cuda'),
          'alpha': torch.tensor(rng.normal(size=(B, S, S)).astype(np.float32), device='cuda'),
          'depth': torch.tensor(rng.normal(size=(B, S, S)).astype(np.float32), device='cuda')}
    o1, gf1, gt1, gl1 = _run(faces, tex, light, fill_back, True, S, upstream=up)
    o0, gf0, gt0, gl0 = _run(faces, tex, light, fill_back, False, S, upstream=up)
    assert torch.equal(o1['alpha'], o0['alpha']) and torch.equal(o1['depth'], o0['depth'])
    assert float(o0['alpha'].sum()) > 50  # the scene draws something
    _close(o1['rgb'], o0['rgb'], LIGHT_ORDER, 'rgb')
    _close(gt1, gt0, GRAD_TOL, 'grad_textures')
    _close(gl1, gl0, GRAD_TOL, 'grad_light')
    _close(gf1, gf0, GRAD_TOL, 'grad_faces')
    assert float(gt0.abs().sum()) > 0 and float(gl0.abs().sum()) > 0


def test_face_light_needs_matching_shapes():
    import neural_renderer_amd as nr
    f = torch.tensor(_faces_scene(1, 8, 1), device='cuda')
    t = torch.rand((1, 8, 2, 2, 2, 3), device='cuda')
    with pytest.raises(ValueError):
        nr.rasterize(f, t, 32, False, face_light=torch.ones((1, 9, 3), device='cuda'))
    with pytest.raises(ValueError):  # neither F nor F / 2 cubes
        nr.rasterize(f, torch.rand((1, 3, 2, 2, 2, 3), device='cuda'), 32, False, face_light=torch.ones((1, 8, 3), device='cuda'))


def test_face_light_only_the_light_gradient():
    """Fixed textures, gradients only through the colours (vertex optimisation of a textured mesh)."""
    B, Nf, ts, S = 2, 40, 4, 64
    rng = np.random.default_rng(9)
    faces = _faces_scene(B, Nf, 11)
    tex = rng.uniform(0, 1, (B, Nf, ts, ts, ts, 3)).astype(np.float32)
    light = rng.uniform(0.2, 1.5, (B, 2 * Nf, 3)).astype(np.float32)
    import neural_renderer_amd as nr
    res = []
    for lit_mode in (True, False):
        f0 = torch.tensor(faces, device='cuda')
        t0 = torch.tensor(tex, device='cuda')
        l0 = torch.tensor(light, device='cuda', requires_grad=True)
        fa = torch.cat((f0, f0.flip(2)), dim=1)
        if lit_mode:
            img = nr.rasterize(fa, t0, S, True, face_light=l0)
        else:
            img = nr.rasterize(fa, torch.cat((t0, t0.permute(0, 1, 4, 3, 2, 5)), dim=1) * l0[:, :, None, None, None, :], S, True)
        (img ** 2).sum().backward()
        res.append((img, l0.grad))
    _close(res[0][0], res[1][0], LIGHT_ORDER, 'rgb')
    _close(res[0][1], res[1][1], GRAD_TOL, 'grad_light')


def _teapot_scene(B, ts, seed):
    rng = np.random.default_rng(seed)
    v, f = H.teapot()
    vb = (v[None] + rng.normal(scale=0.01, size=(B,) + v.shape)).astype(np.float32)
    fb = np.repeat(f[None], B, axis=0)
    tex = rng.uniform(0, 1, (B, f.shape[0], ts, ts, ts, 3)).astype(np.float32)
    eyes = np.array([O.get_points_from_angles(2.732, 20.0 + 5 * i, 70.0 * i) for i in range(B)], np.float32)
    return vb, fb, tex, eyes


@pytest.mark.parametrize('mode,fill_back,per_batch_eye,ts', [('look_at', True, True, 2), ('look_at', True, False, 4),
                                                             ('look', False, True, 3), ('look_at', False, False, 2)])
def test_renderer_face_light_equals_the_default_render(mode, fill_back, per_batch_eye, ts):
    import neural_renderer_amd as nr
    B = 3
    vb, fb, tex, eyes = _teapot_scene(B, ts, 31)
    eye_np = eyes if per_batch_eye else eyes[1]
    rng = np.random.default_rng(32)
    up = torch.tensor(rng.normal(size=(B, 3, 128, 128)).astype(np.float32), device='cuda')
    res = []
    for flag in (True, False):
        r = nr.Renderer()
        r.image_size = 128
        r.camera_mode = mode
        r.fill_back = fill_back
        r.light_direction = [0.3, 0.8, -0.5]
        r.light_color_ambient = [1.0, 0.9, 0.8]
        r.light_color_directional = [0.7, 1.0, 0.6]
        r.light_intensity_ambient = 0.4
        r.light_intensity_directional = 0.6
        if mode == 'look':
            r.camera_direction = [0.2, -0.1, 1.0]
        r.face_light = flag
        v = torch.tensor(vb, device='cuda', requires_grad=True)
        t = torch.tensor(tex, device='cuda', requires_grad=True)
        e = torch.tensor(eye_np, device='cuda', requires_grad=True)
        r.eye = e
        img = r.render(v, torch.tensor(fb, device='cuda'), t)
        assert r.last_frontend == 'fused'
        (img * up).sum().backward()
        res.append((img, v.grad, t.grad, e.grad))
    _close(res[0][0], res[1][0], LIGHT_ORDER, 'images')
    _close(res[0][1], res[1][1], 1e-4, 'grad_vertices')  # float atomics in the face -> vertex scatter (test_frontend_gpu.RTOL)
    _close(res[0][2], res[1][2], GRAD_TOL, 'grad_textures')
    _close(res[0][3], res[1][3], 1e-4, 'grad_eye')


@pytest.mark.parametrize('ts,fill_back', [(2, True), (4, True), (3, False)])
def test_face_light_against_the_oracle(ts, fill_back):
    """Straight against the CPU oracle (not via the lit-texture HIP path): the oracle rasterizes the SAME faces with textures
    lit and duplicated on the host exactly as the reference does (renderer.py:79, lighting.py:50-51); forward within
    LIGHT_ORDER, the gradients (chain rule through the host-side product) within the suite's 1e-4."""
    import neural_renderer_amd as nr
    B, Nf, S = 2, 80, 64
    rng = np.random.default_rng(500 + ts)
    f0 = _faces_scene(B, Nf, 40 + ts, ground=(ts == 4))
    faces = np.concatenate((f0, f0[:, :, ::-1]), axis=1) if fill_back else f0
    tex = rng.uniform(0, 1, (B, Nf, ts, ts, ts, 3)).astype(np.float32)
    light = rng.uniform(0.2, 1.5, (B, faces.shape[1], 3)).astype(np.float32)
    t_all = np.concatenate((tex, tex.transpose((0, 1, 4, 3, 2, 5))), axis=1) if fill_back else tex
    lit = (t_all * light[:, :, None, None, None, :]).astype(np.float32)
    ref = O.rasterize_rgbad(np.ascontiguousarray(faces), lit, S, False, return_function=True)
    g_rgb = rng.normal(size=(B, 3, S, S)).astype(np.float32)
    gf_ref, g_lit = O.rgbad_backward(ref['function'], False, grad_rgb=g_rgb)
    # chain rule of the host-side product: d lit / d tex = light (both copies), d lit / d light = sum over texels
    g_lit = g_lit.astype(np.float64)
    gt_ref = g_lit[:, :Nf] * light[:, :Nf, None, None, None, :]
    if fill_back:
        gt_ref = gt_ref + (g_lit[:, Nf:] * light[:, Nf:, None, None, None, :]).transpose((0, 1, 4, 3, 2, 5))
    gl_ref = (g_lit * t_all).sum(axis=(2, 3, 4))

    ft = torch.tensor(np.ascontiguousarray(faces), device='cuda', requires_grad=True)
    tt = torch.tensor(tex, device='cuda', requires_grad=True)
    lt = torch.tensor(light, device='cuda', requires_grad=True)
    out = nr.rasterize_rgbad(ft, tt, S, False, face_light=lt)
    np.testing.assert_array_equal(out['alpha'].detach().cpu().numpy(), ref['alpha'])
    np.testing.assert_array_equal(out['depth'].detach().cpu().numpy(), ref['depth'])
    assert H.rel_err(out['rgb'].detach().cpu().numpy(), ref['rgb']) <= LIGHT_ORDER
    out['rgb'].backward(torch.tensor(g_rgb, device='cuda'))
    # (RTOL of test_hip_parity: the oracle adds a texel's terms one by one in float, like the reference's atomics; the gathers
    # here add them in another order -- measured 1.5e-5 on these cancelling random cotangents)
    assert H.rel_err(tt.grad.cpu().numpy(), gt_ref) <= 1e-4
    assert H.rel_err(lt.grad.cpu().numpy(), gl_ref) <= 1e-4
    assert H.rel_err(ft.grad.cpu().numpy(), gf_ref) <= 1e-4
