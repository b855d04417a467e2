"""Pin the CPU oracle against every fixture the reference's test-suite holds for the hot path
(SURVEY.md 8c).  CPU only."""
import numpy as np
import pytest

from oracle import oracle as O
import helpers as H


@pytest.fixture(scope='module')
def teapot_batch():
    v, f = H.teapot()
    return v[None], f[None]


def _renderer(**kw):
    r = O.Renderer()
    for k, val in kw.items():
        setattr(r, k, val)
    return r


def test_load_obj_teapot_counts():
    # reference tests/test_load_obj.py:34-37
    v, f = H.teapot()
    assert f.shape[0] == 2464 and v.shape[0] == 1292


def test_silhouette_matches_blender(teapot_batch):
    # reference tests/test_rasterize_silhouettes.py:15-35 (every pixel must match)
    v, f = teapot_batch
    img = _renderer(image_size=256, anti_aliasing=False).render_silhouettes(v, f)[0]
    ref = H.golden()['teapot_blender'].astype(np.float32)
    assert int((img != ref).sum()) == 0
    assert int(img.sum()) == 7580


@pytest.mark.parametrize('S', [64, 256])
def test_unsafe_kernel_restatement_against_the_safe_path_and_the_blender_silhouette(teapot_batch, S):
    """SURVEY 8 row a3' / Appendix B: the reference's "unsafe" visibility kernel K3 (rasterize.py:102-236) has no test of its
    own.  Its sequential emulation (nr_oracle.c: oracle_forward_face_index_map_unsafe) is pinned here the way the survey's
    throw-away restatement was: on the teapot under the default camera it draws exactly the pixels of the safe path (at 256^2:
    of the Blender render the reference ships), picks the same faces, and differs in depth / weights only through its
    x-sorted face_inv -- <= 1.2e-6 / 2.3e-5 at 64^2 and <= 2.6e-6 / 1.0e-4 at 256^2 in the survey."""
    v, f = teapot_batch
    faces = O.Renderer().project(v, f)
    safe = O.Rasterize(S, 0.1, 100, 1e-4, None, False, True, True)
    safe(faces)
    k3 = O.Rasterize(S, 0.1, 100, 1e-4, None, False, True, True)
    k3.unsafe = True
    k3(faces)
    assert int(((safe.face_index_map >= 0) != (k3.face_index_map >= 0)).sum()) == 0
    assert int((safe.face_index_map != k3.face_index_map).sum()) == 0
    assert float(np.abs(safe.depth_map - k3.depth_map).max()) <= (1.3e-6 if S == 64 else 2.7e-6)
    assert float(np.abs(safe.weight_map - k3.weight_map).max()) <= (2.4e-5 if S == 64 else 1.1e-4)
    scale = float(np.abs(safe.face_inv_map).max())
    assert float(np.abs(safe.face_inv_map - k3.face_inv_map).max()) <= 1e-5 * scale  # written back through pi[] (:213-214)
    if S == 256:
        assert int((k3.alpha_map[0][::-1] != H.golden()['teapot_blender'].astype(np.float32)).sum()) == 0


def test_silhouette_batch_of_four_with_empty_slots():
    # reference tests/utils.py:7-24: payload in slot 2, all-zero (degenerate) geometry in slots 0, 1, 3
    v, f = H.teapot()
    vb, fb = H.to_minibatch((v, f))
    # keep the CPU cost down: 128x128 still exercises the degenerate slots
    imgs = _renderer(image_size=128, anti_aliasing=False).render_silhouettes(vb, fb)
    single = _renderer(image_size=128, anti_aliasing=False).render_silhouettes(v[None], f[None])
    assert imgs[[0, 1, 3]].sum() == 0
    np.testing.assert_array_equal(imgs[2], single[0])


def test_depth_matches_blender_and_golden(teapot_batch):
    # reference tests/test_rasterize_depth.py:16-58
    v, f = teapot_batch
    d = _renderer(image_size=256, anti_aliasing=False).render_depth(v, f)[0].copy()
    ref = H.golden()['teapot_blender']
    assert int(((d != d.max()) != ref).sum()) == 0
    d[d == d.max()] = d.min()
    d = (d - d.min()) / (d.max() - d.min())
    np.testing.assert_allclose(d, H.golden()['test_depth'].astype(np.float32) / 255., atol=1e-2)


def test_rgb_ambient_only_matches_blender(teapot_batch):
    # reference tests/test_rasterize.py:52-74
    v, f = teapot_batch
    tex = np.ones((1, f.shape[1], 4, 4, 4, 3), np.float32)
    r = _renderer(image_size=256, anti_aliasing=False, light_intensity_ambient=1.0, light_intensity_directional=0.0)
    img = r.render(v, f, tex)[0].mean(0)
    np.testing.assert_allclose(img, H.golden()['teapot_blender'].astype(np.float32), rtol=1e-4, atol=1e-5)


def test_rgb_golden_png_no_aa(teapot_batch):
    # reference tests/test_rasterize.py:15-32 wrote tests/data/test_rasterize1.png
    v, f = teapot_batch
    tex = np.ones((1, f.shape[1], 4, 4, 4, 3), np.float32)
    img = _renderer(image_size=256, anti_aliasing=False).render(v, f, tex)[0].transpose(1, 2, 0)
    np.testing.assert_array_equal(H.bytescale(img), H.golden()['test_rasterize1'])


def test_rgb_golden_png_aa_other_eye(teapot_batch):
    # reference tests/test_rasterize.py:34-50 wrote tests/data/test_rasterize2.png (AA on, eye [1,1,-2.7])
    v, f = teapot_batch
    tex = np.ones((1, f.shape[1], 4, 4, 4, 3), np.float32)
    img = _renderer(eye=[1, 1, -2.7]).render(v, f, tex)[0].transpose(1, 2, 0)
    np.testing.assert_array_equal(H.bytescale(img), H.golden()['test_rasterize2'])


# ---- known-answer gradients -------------------------------------------------------------------------
# constants restated from reference tests/test_rasterize_silhouettes.py:40-51, :72-83
# (identical in tests/test_rasterize.py:79-90, :116-127)
CASE1 = dict(vertices=[[0.8, 0.8, 1.], [0.0, -0.5, 1.], [0.2, -0.4, 1.]], pyi=25, pxi=35, target=1.0,
             grad_ref=[[1.6725862, -0.26021874, 0.], [1.41986704, -1.64284933, 0.], [0., 0., 0.]])
CASE2 = dict(vertices=[[0.8, 0.8, 1.], [-0.5, -0.8, 1.], [0.8, -0.8, 1.]], pyi=40, pxi=50, target=0.0,
             grad_ref=[[0.98646867, 1.04628897, 0.], [-1.03415668, -0.10403691, 0.], [3.00094461, -1.55173182, 0.]])


def vertex_grad(case, mode):
    """Renderer (image 64, no AA, orthographic) -> loss = |I[py,px] - target| -> d loss / d vertices."""
    v = np.array(case['vertices'], np.float32)[None]
    f = np.array([[0, 1, 2]], np.int32)[None]
    r = _renderer(image_size=64, anti_aliasing=False, perspective=False)
    ff = np.concatenate((f, f[:, :, ::-1]), axis=1)
    faces = O.vertices_to_faces(O.look_at(v, r.eye), ff)
    py, px = case['pyi'], case['pxi']
    if mode == 'alpha':
        out = O.rasterize_rgbad(faces, None, 64, False, return_rgb=False, return_alpha=True, return_depth=False,
                                return_function=True)
        g = np.zeros_like(out['alpha'])
        g[0, py, px] = np.sign(out['alpha'][0, py, px] - case['target'])
        gf, = O.rgbad_backward(out['function'], False, grad_alpha=g)
    else:
        tex = np.ones((1, 2, 4, 4, 4, 3), np.float32)
        out = O.rasterize_rgbad(faces, tex, 64, False, 0.1, 100, 1e-3, [0, 0, 0], True, False, False,
                                return_function=True)
        g = np.zeros_like(out['rgb'])
        g[0, :, py, px] = np.sign(out['rgb'].mean(1)[0, py, px] - case['target']) / 3.0
        gf, _ = O.rgbad_backward(out['function'], False, grad_rgb=g)
    gv = np.zeros((3, 3))
    for fi in range(2):
        for k in range(3):
            gv[ff[0, fi, k]] += gf[0, fi, k]
    # look_at backward (vertices' = (v - eye) R^T  =>  dv = dv' R); R ~ identity * (1 - 4e-6)
    eye = np.array(r.eye, np.float32)
    z = O._normalize((-eye)[None])
    x = O._normalize(np.cross(np.array([[0, 1, 0]], np.float32), z).astype(np.float32))
    y = O._normalize(np.cross(z, x).astype(np.float32))
    R = np.concatenate((x, y, z), 0).astype(np.float64)
    return gv @ R


@pytest.mark.parametrize('case', [CASE1, CASE2], ids=['out_of_face', 'on_face'])
def test_backward_silhouette_grad_ref(case):
    # reference tests/test_rasterize_silhouettes.py:37-99 (rtol 1e-2); the oracle is within 1e-5
    np.testing.assert_allclose(vertex_grad(case, 'alpha'), np.array(case['grad_ref']), rtol=1e-5, atol=1e-7)


@pytest.mark.parametrize('case', [CASE1, CASE2], ids=['out_of_face', 'on_face'])
def test_backward_rgb_grad_ref(case):
    # reference tests/test_rasterize.py:76-149 (rtol 1e-2, eps 1e-3 path)
    np.testing.assert_allclose(vertex_grad(case, 'rgb'), np.array(case['grad_ref']), rtol=1e-2, atol=1e-7)


def test_backward_depth_finite_differences():
    """The intent of reference tests/test_rasterize_depth.py:60-93 done properly (that test is vacuous:
    it reads batch slot 0 while the data is in slot 2): analytic K8 gradient vs forward differences."""
    vertices = np.array([[-0.9, -0.9, 2.], [-0.8, 0.8, 1.], [0.8, 0.8, 0.5]], np.float32)
    faces_i = np.array([[0, 1, 2], [2, 1, 0]], np.int32)  # fill_back (renderer.py:58)
    py, px = 15, 20

    def depth_at(v):
        faces = O.vertices_to_faces(v[None], faces_i[None])
        out = O.rasterize_rgbad(faces, None, 64, False, return_rgb=False, return_alpha=False, return_depth=True,
                                return_function=True)
        return out

    out = depth_at(vertices)
    d0 = float(out['depth'][0, py, px])
    assert d0 < 100  # the probe pixel is on the face
    g = np.zeros_like(out['depth'])
    g[0, py, px] = 2 * (d0 - 1)
    gf, = O.rgbad_backward(out['function'], False, grad_depth=g)
    grad = gf[0, 0] + gf[0, 1][::-1]
    grad2 = np.zeros((3, 3))
    h = 1e-3
    for i in range(3):
        for j in range(3):
            v2 = vertices.copy()
            v2[i, j] += h
            d2 = float(depth_at(v2)['depth'][0, py, px])
            grad2[i, j] = ((d2 - 1) ** 2 - (d0 - 1) ** 2) / h
    np.testing.assert_allclose(grad, grad2, atol=2e-3)
