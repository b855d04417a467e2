"""CPU ORACLE #2 (test infrastructure, NOT product code) -- the "naive NumPy per-pixel loop" of BASELINE.json
configs[0] (teapot, 1 view, 64x64 silhouette, no GPU).

A second, independently structured restatement of the reference's hot path
(/root/reference/neural_renderer/rasterize.py), written with NumPy float32 array arithmetic instead of the scalar C of
oracle/nr_oracle.c:

  * forward_face_index_map: one Python iteration per PIXEL, all faces of the image tested at once   (K1 :240-277, K2 :279-359)
  * forward_texture_sampling / background / alpha: per covered pixel                                (K4 :361-438, K5 :440-465)
  * backward_pixel_map: one Python iteration per (edge, axis, line d0), all faces and the whole line d1 at once (K6 :517-748)

Two restatements that share no code and no loop structure agreeing bit for bit (integers, float maps) and to summation
round-off (gradients) is the cross-check tests/test_numpy_naive.py performs; the C oracle in turn is pinned against the
reference's golden fixtures (tests/test_oracle_golden.py).  Every float32 operation below is a separately rounded NumPy
float32 operation in the reference's order; expressions the CUDA text evaluates in double (its `0.5`, `2.`, `1.`, pasted
near / far / eps literals) are evaluated in float64 and rounded once.

Only tests/ may import this module (it is far too slow for anything else: ~10 s for configs[0]).
"""
import numpy as np

f32 = np.float32
f64 = np.float64


def f2i(x):
    """CUDA `(int)x`: truncate toward zero, saturate, NaN -> 0 (int64 so that `+ direction` cannot wrap)."""
    x = np.asarray(x, f64)
    return np.where(np.isnan(x), 0.0, np.clip(x, -2147483648.0, 2147483647.0)).astype(np.int64)


def is_backside(f):
    """rasterize.py:252 / :306 / :540 on faces [F,3,3]."""
    return (f[:, 2, 1] - f[:, 0, 1]) * (f[:, 1, 0] - f[:, 0, 0]) < (f[:, 1, 1] - f[:, 0, 1]) * (f[:, 2, 0] - f[:, 0, 0])


def to_pixel(xy, s):
    """rasterize.py:258 / :549: p = 0.5 * (x * is + is - 1), the 0.5 being a double literal."""
    t = xy * f32(s) + f32(s) - f32(1)
    return (0.5 * t.astype(f64)).astype(f32)


def face_inv(f, s):
    """K1 (rasterize.py:240-277) for the faces [F,3,3] of one image; zeros for back faces."""
    p = to_pixel(f[:, :, :2], s)
    p0x, p0y, p1x, p1y, p2x, p2y = p[:, 0, 0], p[:, 0, 1], p[:, 1, 0], p[:, 1, 1], p[:, 2, 0], p[:, 2, 1]
    with np.errstate(all='ignore'):
        m = np.stack([
            p1y - p2y, p2x - p1x, p1x * p2y - p2x * p1y,
            p2y - p0y, p0x - p2x, p2x * p0y - p0x * p2y,
            p0y - p1y, p1x - p0x, p0x * p1y - p1x * p0y], axis=1)                   # :261-264
        den = p2x * (p0y - p1y) + p0x * (p1y - p2y) + p1x * (p2y - p0y)             # :265-268
        m = m / den[:, None]                                                        # :269
    m[is_backside(f)] = 0                                                           # :252 (zeros_like, :240)
    return m.reshape(-1, 3, 3).astype(f32)


def forward_face_index_map(faces, s, near, far, return_face_inv=False):
    """K1 + K2, per-pixel loop.  Returns face_index_map, weight_map, depth_map[, face_inv_map] initialised as :478-496."""
    faces = np.ascontiguousarray(faces, f32)
    bs, nf = faces.shape[:2]
    fi = np.full((bs, s, s), -1, np.int32)
    weight = np.zeros((bs, s, s, 3), f32)
    depth = np.zeros((bs, s, s), f32) + f32(far)
    inv_map = np.zeros((bs, s, s, 3, 3), f32) if return_face_inv else None
    with np.errstate(all='ignore'):
        for bn in range(bs):
            f = faces[bn]
            inv = face_inv(f, s)
            front = ~is_backside(f)
            x0, y0, z0 = f[:, 0, 0], f[:, 0, 1], f[:, 0, 2]
            x1, y1, z1 = f[:, 1, 0], f[:, 1, 1], f[:, 1, 2]
            x2, y2, z2 = f[:, 2, 0], f[:, 2, 1], f[:, 2, 2]
            ex0, ey0 = x1 - x0, y1 - y0
            ex1, ey1 = x2 - x1, y2 - y1
            ex2, ey2 = x0 - x2, y0 - y2
            for yi in range(s):
                yp = f32((2. * yi + 1 - s) / s)                                     # :291
                for xi in range(s):
                    xp = f32((2. * xi + 1 - s) / s)                                 # :292
                    out = (((yp - y0) * ex0 < (xp - x0) * ey0) |                    # :310-312
                           ((yp - y1) * ex1 < (xp - x1) * ey1) |
                           ((yp - y2) * ex2 < (xp - x2) * ey2))
                    cand = np.nonzero(front & ~out)[0]
                    if cand.size == 0:
                        continue
                    m = inv[cand]
                    w = m[:, :, 0] * f32(xi) + m[:, :, 1] * f32(yi) + m[:, :, 2]    # :317-319
                    w = np.fmin(np.fmax(w, f32(0)), f32(1))                         # :323 (NaN-ignoring min/max)
                    w_sum = ((f32(0) + w[:, 0]) + w[:, 1]) + w[:, 2]                # :324
                    w = w / w_sum[:, None]                                          # :326-327
                    t = w[:, 0] / z0[cand] + w[:, 1] / z1[cand] + w[:, 2] / z2[cand]
                    zp = (1. / t.astype(f64)).astype(f32)                           # :330
                    zd = zp.astype(f64)
                    skip = (zd <= near) | (far <= zd)                               # :331
                    depth_min = f32(far)
                    best = -1
                    for j in range(cand.size):                                      # ascending fn, strict < : :334
                        if (not skip[j]) and zp[j] < depth_min:
                            depth_min = zp[j]
                            best = j
                    if best >= 0:                                                   # :343-348
                        fi[bn, yi, xi] = cand[best]
                        weight[bn, yi, xi] = w[best]
                        depth[bn, yi, xi] = depth_min
                        if return_face_inv:
                            inv_map[bn, yi, xi] = m[best]
    if return_face_inv:
        return fi, weight, depth, inv_map
    return fi, weight, depth


def forward_texture_sampling(faces, textures, fi, weight, depth, eps, background=(0, 0, 0)):
    """K4 + K5 rgb (rasterize.py:361-438, :451-465), per covered pixel.  Literal Q1: face z from batch 0 (:389)."""
    faces = np.ascontiguousarray(faces, f32)
    textures = np.ascontiguousarray(textures, f32)
    bs, s = fi.shape[:2]
    ts = textures.shape[2]
    tex = textures.reshape(bs, -1, ts * ts * ts, 3)
    bg = np.broadcast_to(np.asarray(background, f32), (bs, 3))
    rgb = np.zeros((bs, s, s, 3), f32)
    hi = float(ts - 1) - float(eps)
    for bn, yi, xi in zip(*np.nonzero(fi >= 0)):
        fn = fi[bn, yi, xi]
        z = faces[0, fn, :, 2]                                                       # :389
        tif = weight[bn, yi, xi] * f32(ts - 1) * (depth[bn, yi, xi] / z)             # :399
        tif = np.fmax(tif.astype(f64), 0.).astype(f32)                               # :400
        tif = np.fmin(tif.astype(f64), hi).astype(f32)                               # :401
        ti = f2i(tif)
        frac = tif - ti.astype(f32)
        pix = np.zeros(3, f32)
        for pn in range(8):                                                          # :407-426
            w = f32(1)
            idx = [0, 0, 0]
            for k in range(3):
                if (pn >> k) % 2 == 0:
                    w = w * (f32(1) - frac[k])
                    idx[k] = int(ti[k])
                else:
                    w = w * frac[k]
                    idx[k] = int(ti[k]) + 1
            isc = idx[0] * ts * ts + idx[1] * ts + idx[2]
            pix = pix + w * tex[bn, fn, isc]
        rgb[bn, yi, xi] = pix
    mask = (fi >= 0).astype(f32)[..., None]                                          # :461
    return rgb * mask + (f32(1) - mask) * bg[:, None, None, :]                       # :463-465


def forward_alpha_map(fi):
    """K5 alpha (rasterize.py:440-449)."""
    return (fi >= 0).astype(f32)


def _dist(p0x, p1x, den, d1, d1_cross, s, eps):
    """rasterize.py:649-651 / :720-722: signed distance in NDC with the +-eps guard (double literals `2.` and eps)."""
    d = ((p1x - p0x) / den)[:, None] * (d1[None, :].astype(f32) - d1_cross[:, None])
    d = (d.astype(f64) * 2. / s).astype(f32)
    return np.where(0 < d, d.astype(f64) + eps, d.astype(f64) - eps).astype(f32)


def backward_pixel_map(faces, fi, rgb, alpha, g_rgb, g_alpha, eps):
    """K6 (rasterize.py:517-748).  rgb / alpha (and their gradients) may be None.  Every term is the reference's float32
    arithmetic; the per-face sums are accumulated in float64 and rounded once (compare with the C oracle's
    accumulate_double form)."""
    faces = np.ascontiguousarray(faces, f32)
    bs, nf = faces.shape[:2]
    s = fi.shape[1]
    chans, grads = [], []
    if alpha is not None:                                                            # alpha first (:631-633), then r,g,b
        chans.append(np.asarray(alpha, f32)[..., None])
        grads.append(np.asarray(g_alpha, f32)[..., None])
    if rgb is not None:
        chans.append(np.asarray(rgb, f32))
        grads.append(np.asarray(g_rgb, f32))
    img = np.concatenate(chans, axis=-1)
    gim = np.concatenate(grads, axis=-1)
    nc = img.shape[-1]
    grad = np.zeros((bs, nf, 3, 3), f64)
    grid = np.arange(s, dtype=np.int64)

    def channel_diff(line, gline, ref):
        d = np.zeros((ref.shape[0], s), f32)
        for c in range(nc):
            d = d + (line[None, :, c] - ref[:, None, c]) * gline[None, :, c]
        return d

    with np.errstate(all='ignore'):
        for bn in range(bs):
            f = faces[bn]
            front = ~is_backside(f)                                                  # :540
            for e in range(3):
                pi = [(e + k) % 3 for k in range(3)]                                 # :547
                pp = to_pixel(f[:, pi, :2], s)                                       # :549
                for axis in range(2):
                    order = [axis, 1 - axis]                                         # :555 p[num][dim] = pp[num][(dim+axis)%2]
                    p0x, p0y = pp[:, 0, order[0]], pp[:, 0, order[1]]
                    p1x, p1y = pp[:, 1, order[0]], pp[:, 1, order[1]]
                    p2x, p2y = pp[:, 2, order[0]], pp[:, 2, order[1]]
                    lt = p0x < p1x
                    direction = np.where(lt, -1, 1) if axis == 0 else np.where(lt, 1, -1)   # :559-564
                    d0_from = f2i(np.fmax(np.ceil(np.fmin(p0x, p1x)).astype(f64), 0.))     # :568
                    d0_to = f2i(np.fmin(np.fmax(p0x, p1x).astype(f64), s - 1.))            # :569
                    slope = (p1y - p0y) / (p1x - p0x)
                    for d0 in range(s):
                        sel = np.nonzero(front & (d0_from <= d0) & (d0 <= d0_to))[0]
                        if sel.size == 0:
                            continue
                        fd0 = f32(d0)
                        d1_cross = slope[sel] * (fd0 - p0x[sel]) + p0y[sel]          # :573
                        dr = direction[sel]
                        d1_in = np.where(0 < dr, f2i(np.floor(d1_cross)), f2i(np.ceil(d1_cross)))   # :574
                        d1_out = d1_in + dr                                          # :575
                        ok = (0 <= d1_in) & (d1_in < s) & (0 <= d1_out) & (d1_out < s)   # :578-579
                        sel, d1_cross, dr, d1_in, d1_out = sel[ok], d1_cross[ok], dr[ok], d1_in[ok], d1_out[ok]
                        if sel.size == 0:
                            continue
                        a0x, a1x, a2x = p0x[sel], p1x[sel], p2x[sel]
                        a0y, a1y, a2y = p0y[sel], p1y[sel], p2y[sel]
                        if axis == 0:                                                # :587-593 line d0 = column x
                            line, gline, fline = img[bn, :, d0], gim[bn, :, d0], fi[bn, :, d0]
                        else:
                            line, gline, fline = img[bn, d0, :], gim[bn, d0, :], fi[bn, d0, :]

                        # out sweep, :604-659
                        lim = np.where(0 < dr, s - 1, 0)
                        lo = np.maximum(np.minimum(d1_out, lim), 0)
                        hi = np.minimum(np.maximum(d1_out, lim), s - 1)
                        diff_out = channel_diff(line, gline, line[d1_in])
                        m_out = ((fline[d1_in] == sel)[:, None] & (lo[:, None] <= grid) & (grid <= hi[:, None])
                                 & ~(diff_out <= 0))                                 # :605, :647

                        # in sweep, :662-730
                        between = (fd0 - a0x) * (fd0 - a2x) < 0                      # :665
                        c_a = (a2y - a0y) / (a2x - a0x) * (fd0 - a0x) + a0y
                        c_b = (a1y - a2y) / (a1x - a2x) * (fd0 - a2x) + a2y
                        cross2 = np.where(between, c_a, c_b)
                        lim2 = np.where(0 < dr, f2i(np.ceil(cross2)), f2i(np.floor(cross2)))   # :671-672
                        lo2 = np.maximum(np.minimum(d1_in, lim2), 0)
                        hi2 = np.minimum(np.maximum(d1_in, lim2), s - 1)
                        diff_in = channel_diff(line, gline, line[d1_out])
                        m_in = ((fline[None, :] == sel[:, None]) & (lo2[:, None] <= grid) & (grid <= hi2[:, None])
                                & ~(diff_in <= 0))                                   # :707, :717

                        for mask, diff in ((m_out, diff_out), (m_in, diff_in)):
                            if not mask.any():
                                continue
                            # vertex pi[0]: :648-652 / :719-723 ; vertex pi[1]: :653-657 / :724-728
                            for vert, den, on in ((pi[0], a1x - fd0, a1x != fd0), (pi[1], fd0 - a0x, a0x != fd0)):
                                dist = _dist(a0x, a1x, den, grid, d1_cross, s, eps)
                                term = np.where(mask & on[:, None], (diff / dist).astype(f64), 0.0)
                                np.subtract.at(grad[bn, :, vert, 1 - axis], sel, term.sum(axis=1))
    out = grad.astype(f32)
    out[:, :, :, 2] = 0
    return out
