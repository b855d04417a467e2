"""The error bar on "bit-exact vs the reference" (CPU, oracle only; DESIGN.md 3 "Contraction").

The oracle and the HIP kernels share the UN-fused reading of the reference's CUDA text.  nvcc's default (-fmad=true) may
contract the `a*b+c` of rasterize.py:258 (pixel coordinates), :261-269 (inverse barycentric matrix) and :317-319 (weights)
into fused multiply-adds; the reference cannot be run here, so the oracle carries a switch that evaluates exactly those
expressions fused (oracle.set_contraction), and this test pins what moves on the reference's own teapot views:

  * the coverage tests (:252 / :306, :310-312) compare products and contain no add to fuse with: no pixel changes between
    covered and uncovered, and on this scene no pixel changes owner either (profiles/r03_contraction_study.jsonl: 0 of
    4.2 M pixels on the headline batch, 7 on config 4, 71 of 1 M on config 5 -- near-ties in depth between faces that overlap
    at a pixel centre);
  * weights and depth DO move, and by far more than an ulp: the cofactors `p1x*p2y - p2x*p1y` cancel catastrophically for
    small faces (products ~1e4, difference ~1e1), so fusing one product changes face_inv[.,2] at the 1e-4 level and the
    weights of tiny faces by up to 2e-2 (median 9e-6).  Both evaluations are "the reference's arithmetic"; which one an
    NVIDIA build produces depends on its compiler.  Parity here means: identical to the un-fused evaluation, bit for bit.
"""
import numpy as np

from oracle import oracle as O
import helpers as H


def _maps(faces, S, contract):
    O.set_contraction(contract)
    try:
        fn = O.Rasterize(S, 0.1, 100, 1e-3, (0, 0, 0), False, True, True)
        fn.blocked = True
        fn(faces)
    finally:
        O.set_contraction(False)
    return fn.face_index_map, fn.depth_map, fn.weight_map


def test_fused_multiply_add_reading_of_k1_k2():
    assert O.get_contraction() is False  # the parity convention is the default
    faces, _ = H.teapot_views(8, 256)
    fi0, d0, w0 = _maps(faces, 256, False)
    fi1, d1, w1 = _maps(faces, 256, True)
    assert O.get_contraction() is False
    # again un-fused: bit-identical (the switch leaves nothing behind)
    fi2, d2, w2 = _maps(faces, 256, False)
    assert np.array_equal(fi0, fi2) and np.array_equal(d0, d2) and np.array_equal(w0, w2)
    cov = fi0 >= 0
    assert 50000 < int(cov.sum()) < 80000
    # ownership: pinned count
    assert int((fi0 != fi1).sum()) == 0
    dw = np.abs(w1[cov].astype(np.float64) - w0[cov]).max(axis=-1)
    dd = np.abs(d1[cov].astype(np.float64) - d0[cov]) / np.abs(d0[cov])
    assert (dw > 0).mean() > 0.9            # the switch does switch: nearly every covered pixel's weights move ...
    assert 1e-3 < dw.max() < 5e-2           # ... by up to ~2e-2 for the tiniest faces (cancellation in the cofactors)
    assert np.median(dw) < 5e-5 and np.percentile(dw, 99) < 2e-3
    assert dd.max() < 1e-3 and np.median(dd) < 1e-6


def test_contraction_switch_leaves_coverage_tests_alone():
    """Random soup incl. degenerate faces: the set of covered pixels is identical under both readings."""
    rng = np.random.default_rng(5)
    faces = H.random_scene(rng, 2, 300)
    fi0, _, _ = _maps(faces, 96, False)
    fi1, _, _ = _maps(faces, 96, True)
    assert np.array_equal(fi0 >= 0, fi1 >= 0)
    assert int((fi0 != fi1).sum()) <= 2  # overlapping random faces: an owner may flip at a depth near-tie
