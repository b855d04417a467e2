"""Masked Adam (reference optimizers.py:9-39) and Mesh.  No reference test exists ("parity unpinned"): the
oracle is the literal float32 restatement; the CPU test checks the torch formulation against it, the GPU test the HIP
kernel (bit for bit)."""
import numpy as np
import pytest
import torch

from oracle import oracle as O


def _run(device, steps=5):
    import neural_renderer_amd as nr
    rng = np.random.default_rng(7)
    p0 = [rng.normal(size=(50, 3)).astype(np.float32), rng.normal(size=(7, 2, 2, 2, 3)).astype(np.float32)]
    params = [torch.nn.Parameter(torch.tensor(p, device=device)) for p in p0]
    params[1].lr = 0.25                                     # per-parameter multiplier (Mesh.set_lr)
    opt = nr.Adam(params, alpha=0.01, beta1=0.5)            # the examples' settings
    ref = [p.copy() for p in p0]
    oracle = O.Adam(alpha=0.01, beta1=0.5)
    for _ in range(steps):
        grads = [rng.normal(size=p.shape).astype(np.float32) for p in p0]
        for g in grads:
            g[rng.uniform(size=g.shape) < 0.4] = 0          # elements no pixel saw
        for p, g in zip(params, grads):
            p.grad = torch.tensor(g, device=device)
        opt.step()
        oracle.update(ref, grads, lr_mult=[1.0, 0.25])
    return [p.detach().cpu().numpy() for p in params], ref, p0, grads


def test_masked_adam_torch_formulation_matches_oracle():
    got, ref, p0, grads = _run('cpu')
    for a, b in zip(got, ref):
        np.testing.assert_allclose(a, b, rtol=1e-6, atol=1e-7)
    # an element whose gradient was zero in every step never moves; with one parameter's lr = 0 it is skipped entirely
    import neural_renderer_amd as nr
    p = torch.nn.Parameter(torch.ones(4))
    p.grad = torch.tensor([0.0, 1.0, 0.0, -2.0])
    p2 = torch.nn.Parameter(torch.ones(4))
    p2.lr = 0
    p2.grad = torch.ones(4)
    nr.Adam([p, p2], alpha=0.1).step()
    assert p[0] == 1 and p[2] == 1 and p[1] < 1 and p[3] > 1 and torch.all(p2 == 1)


@pytest.mark.gpu
def test_masked_adam_hip_kernel_bit_exact():
    got, ref, _, _ = _run('cuda')
    for a, b in zip(got, ref):
        np.testing.assert_array_equal(a, b)


def test_mesh_module(tmp_path):
    import neural_renderer_amd as nr
    path = str(tmp_path / 't.obj')
    with open(path, 'w') as f:
        f.write('v 1 0 0\nv 0 1 0\nv 0 0 1\nv 0 0 0\nf 2 4 3\nf 4 2 1\nf 3 1 2\nf 1 3 4\n')   # the reference's tetrahedron
    mesh = nr.Mesh(path, texture_size=3)
    assert mesh.num_vertices == 4 and mesh.num_faces == 4 and mesh.textures.shape == (4, 3, 3, 3, 3)
    assert float(mesh.textures.detach().std()) < 0.1               # chainer.initializers.Normal(): scale 0.05 (mesh.py:22)
    v, f, t = mesh.get_batch(5)
    assert v.shape == (5, 4, 3) and f.shape == (5, 4, 3) and t.shape == (5, 4, 3, 3, 3, 3)
    assert float(t.min()) > 0 and float(t.max()) < 1      # sigmoid, mesh.py:33
    mesh.set_lr(0.5, 2.0)
    assert mesh.vertices.lr == 0.5 and mesh.textures.lr == 2.0
