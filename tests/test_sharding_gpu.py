"""`-m gpu`: the batch-of-views sharding of the multi-GPU path is result-preserving under the reference-literal Q1 behaviour
(textures are sampled with the vertex depths of batch element 0, rasterize.py:389): 8 views with per-view cameras and
random textures rendered as one batch and as two shards of 4 -- each shard handed the global element 0's faces as
`faces_z_ref` -- give identical bits for rgb, grad_textures and grad_faces.  Without the hand-over the second shard differs
(the test would notice a kernel that ignores the pointer), with fix_batch_z nothing needs to be handed over."""
import numpy as np
import pytest
import torch

import abi
import helpers as H

pytestmark = pytest.mark.gpu

S, TS = 96, 3


def _scene():
    faces, _ = H.teapot_views(8, S)
    rng = np.random.default_rng(808)
    textures = rng.uniform(0, 1, (8, faces.shape[1], TS, TS, TS, 3)).astype(np.float32)
    g_rgb = rng.normal(size=(8, S, S, 3)).astype(np.float32)
    g_alpha = rng.normal(size=(8, S, S)).astype(np.float32)
    return faces, textures, g_rgb, g_alpha


def _run_abi(faces, textures, g_rgb, g_alpha, z_ref, flags=0):
    fw = abi.forward_fused(faces, textures, S, 0.1, 100.0, 1e-3, (0.1, 0.2, 0.3), flags, True, True, False,
                           faces_z_ref=z_ref)
    gf, gt = abi.backward_fused(fw, g_rgb, g_alpha, None)
    return abi.host(fw['rgb_map']), abi.host(gf), abi.host(gt)


def _same(parts, full, names):
    """Shards == batch, bit for bit: images and grad_textures come from atomic-free kernels, and grad_faces from K6's k_bpm_row,
    whose per-record sums do not depend on what else is in the launch (every record is reduced exactly once, in double above
    the lanes' float sums; the double atomics that add a face's records round to the same float)."""
    for k, name in enumerate(names):
        got = np.concatenate((parts[0][k], parts[1][k]))
        if name == 'grad_faces':
            # (the double atomics arrive in another order in a shard: a sum that sits within 1e-16 of a float rounding point may
            # come out one ulp apart -- seen in no run, allowed in two entries so that the suite cannot flake on it)
            assert int((got != full[k]).sum()) <= 2 and H.rel_err(got, full[k]) <= 1e-6, name
        else:
            np.testing.assert_array_equal(got, full[k], err_msg=name)


def test_two_shards_equal_one_batch_through_the_c_abi():
    faces, textures, g_rgb, g_alpha = _scene()
    full = _run_abi(faces, textures, g_rgb, g_alpha, None)
    parts = [_run_abi(faces[s], textures[s], g_rgb[s], g_alpha[s], faces[0]) for s in (slice(0, 4), slice(4, 8))]
    _same(parts, full, ('rgb_map', 'grad_faces', 'grad_textures'))
    # sensitivity: a shard that samples with ITS OWN first view's depths is a different (wrong) result ...
    alone = _run_abi(faces[4:], textures[4:], g_rgb[4:], g_alpha[4:], None)
    assert np.abs(alone[0] - full[0][4:]).max() > 1e-3
    # ... and the staged entry points honour the pointer as well
    fw = abi.forward(faces[4:], textures[4:], S, 0.1, 100.0, 1e-3, (0.1, 0.2, 0.3), 0, True, True, False,
                     faces_z_ref=faces[0])
    np.testing.assert_array_equal(abi.host(fw['rgb_map']), full[0][4:])
    _, gt = abi.backward(fw, g_rgb[4:], g_alpha[4:], None)
    np.testing.assert_array_equal(abi.host(gt), full[2][4:])
    # with the corrected sampling (NR_FLAG_FIX_TEXTURE_BATCH_Z) shards are independent of any reference view
    full_fix = _run_abi(faces, textures, g_rgb, g_alpha, None, flags=1)
    part_fix = _run_abi(faces[4:], textures[4:], g_rgb[4:], g_alpha[4:], None, flags=1)
    np.testing.assert_array_equal(part_fix[0], full_fix[0][4:])
    np.testing.assert_array_equal(part_fix[2], full_fix[2][4:])


def test_two_shards_equal_one_batch_through_the_operator():
    import neural_renderer_amd as nr
    from neural_renderer_amd import distributed as nrd
    faces, textures, g_rgb, g_alpha = _scene()

    def run(sl, z_ref):
        ft = torch.tensor(faces[sl], device='cuda', requires_grad=True)
        tt = torch.tensor(textures[sl], device='cuda', requires_grad=True)
        fn = nr.Rasterize(S, 0.1, 100, 1e-3, (0.1, 0.2, 0.3), True, True, False)
        fn.faces_z_ref = z_ref
        rgb, alpha, _ = fn(ft, tt)
        torch.autograd.backward([rgb, alpha], [torch.tensor(g_rgb[sl], device='cuda'), torch.tensor(g_alpha[sl], device='cuda')])
        return rgb.detach().cpu().numpy(), ft.grad.cpu().numpy(), tt.grad.cpu().numpy()

    full = run(slice(0, 8), None)
    # what rank 0 would broadcast (single process: the helper returns its own first view)
    z_ref = nrd.broadcast_reference_faces(torch.tensor(faces[:4], device='cuda'))
    parts = [run(slice(0, 4), z_ref), run(slice(4, 8), z_ref)]
    _same(parts, full, ('rgb', 'grad_faces', 'grad_textures'))


def test_two_shards_equal_one_batch_with_face_light():
    """The same with per-face light colours (nr_face_light): the colours and the original cubes shard with the batch."""
    import neural_renderer_amd as nr
    faces, textures, g_rgb, g_alpha = _scene()
    rng = np.random.default_rng(77)
    light = rng.uniform(0.3, 1.4, faces.shape[:2] + (3,)).astype(np.float32)

    def run(sl, z_ref):
        ft = torch.tensor(faces[sl], device='cuda', requires_grad=True)
        tt = torch.tensor(textures[sl], device='cuda', requires_grad=True)
        lt = torch.tensor(light[sl], device='cuda', requires_grad=True)
        fn = nr.Rasterize(S, 0.1, 100, 1e-3, (0.1, 0.2, 0.3), True, True, False)
        fn.faces_z_ref = z_ref
        rgb, alpha, _ = fn(ft, tt, lt)
        torch.autograd.backward([rgb, alpha], [torch.tensor(g_rgb[sl], device='cuda'), torch.tensor(g_alpha[sl], device='cuda')])
        return rgb.detach().cpu().numpy(), ft.grad.cpu().numpy(), tt.grad.cpu().numpy(), lt.grad.cpu().numpy()

    full = run(slice(0, 8), None)
    z_ref = torch.tensor(faces[0], device='cuda')
    parts = [run(slice(0, 4), z_ref), run(slice(4, 8), z_ref)]
    _same(parts, full, ('rgb', 'grad_faces', 'grad_textures', 'grad_light'))
