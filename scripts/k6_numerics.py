"""K6 numerics study (development / profiles/r04_k6_numerics.jsonl): what each arithmetic knob of the default kernel
(csrc/nr_k6_tune.h) costs in time and buys in accuracy, on the BASELINE configurations at full size.

For every library given (variant builds of neural_renderer_amd._build.build_variant; '' = the product library) the stage
call nr_backward_pixel_map runs on the same residual maps and upstream gradients; its grad_faces are compared with the
oracle's K6 terms summed in double (the exactly summed reference terms), in the floor metric of the parity tests
(tests/helpers.rel_err) and elementwise, and the stage is timed with HIP events.

    VARIANTS="newton bd" SCENES="H C4 C5" python scripts/k6_numerics.py > gpurun_out/k6_numerics.jsonl
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import torch

from neural_renderer_amd import _lib, _build
from oracle import oracle as O
import abi
import helpers as H


def scene(name):
    if name == 'H':
        faces, _ = H.teapot_views(64, 256)
        rng = np.random.default_rng(640)
        tex = rng.uniform(0, 1, (64, faces.shape[1], 2, 2, 2, 3)).astype(np.float32)
        return faces, tex, 256, (True, True), (0.1, 0.2, 0.3), 641
    if name == 'C4':
        from test_full_size_gpu import config4_meshes
        faces = config4_meshes(64)
        rng = np.random.default_rng(44)
        tex = rng.uniform(0, 1, (64, faces.shape[1], 4, 4, 4, 3)).astype(np.float32)
        return faces, tex, 256, (True, False), (0.0, 0.0, 0.0), 45
    if name == 'C5':
        from test_hip_parity import icosphere, project_mesh
        rng = np.random.default_rng(55)
        v0, f0 = icosphere(7)
        v = v0 * (0.6 + 0.02 * rng.normal(size=(v0.shape[0], 1))).astype(np.float32)
        faces = project_mesh(v.astype(np.float32), f0, [0.0, 0.0, -2.4])[None]
        tex = rng.uniform(0, 1, (1, faces.shape[1], 2, 2, 2, 3)).astype(np.float32)  # (K6 does not see the texture size)
        return faces, tex, 1024, (True, True), (0.0, 0.0, 0.0), 56
    if name == 'X3':  # 4 teapot views at 1024^2
        faces, _ = H.teapot_views(4, 1024)
        tex = np.ones((4, faces.shape[1], 2, 2, 2, 3), np.float32)
        return faces, tex, 1024, (True, True), (0.0, 0.0, 0.0), 7
    if name == 'X4':  # 256 views at 128^2
        faces, _ = H.teapot_views(256, 128)
        tex = np.ones((256, faces.shape[1], 2, 2, 2, 3), np.float32)
        return faces, tex, 128, (True, True), (0.0, 0.0, 0.0), 8
    raise KeyError(name)


def use_library(tag):
    _lib._lib = None
    if tag:
        os.environ['NR_HIP_LIB'] = os.path.join(os.path.dirname(_build.LIB_PATH), 'libnr_hip_%s.so' % tag)
    else:
        os.environ.pop('NR_HIP_LIB', None)
    return _lib.load()


def k6_stage(lib, fw, gr, ga, flags, iters):
    B, F, S = fw['B'], fw['F'], fw['S']
    gf = torch.full((B, F, 3, 3), float('nan'), device='cuda')
    wsb = lib.nr_backward_workspace_bytes(B, F, S, int(gr is not None), int(ga is not None))
    ws = torch.empty(max(wsb, 1), dtype=torch.uint8, device='cuda')
    st = torch.cuda.current_stream().cuda_stream

    def call():
        return lib.nr_backward_pixel_map(
            fw['faces'].data_ptr(), fw['face_index_map'].data_ptr(), _lib.ptr(fw.get('rgb_map')) if gr is not None else None,
            _lib.ptr(fw.get('alpha_map')) if ga is not None else None, _lib.ptr(gr), _lib.ptr(ga), gf.data_ptr(), B, F, S,
            fw['eps'], int(gr is not None), int(ga is not None), flags, _lib.ptr(fw.get('visible_faces')), ws.data_ptr(), wsb, st)
    _lib.check(call(), 'k6')
    torch.cuda.synchronize()
    out = gf.cpu().numpy()
    us = None
    if iters:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(3):
            call()
        e0.record()
        for _ in range(iters):
            call()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / iters
    return out, us


def main():
    O.build()
    variants = [''] + os.environ.get('VARIANTS', '').split()
    iters = int(os.environ.get('ITERS', 20))
    for name in os.environ.get('SCENES', 'H C4').split():
        faces, tex, S, (rgb, alpha), bg, seed = scene(name)
        t0 = time.time()
        eps = float(os.environ.get('EPS', 1e-3))
        fn = O.Rasterize(S, 0.1, 100, eps, bg, rgb, alpha, False)
        fn.blocked = True
        fn(faces, tex) if rgb else fn(faces)
        rng = np.random.default_rng(seed)
        shape = fn.face_index_map.shape
        g_rgb = rng.normal(size=shape + (3,)).astype(np.float32) if rgb else None
        g_alpha = rng.normal(size=shape).astype(np.float32) if alpha else None
        ref = fn.backward(g_rgb, g_alpha, None, accumulate_double=True, skip_textures=True)[0]
        noise = H.rel_err(fn.backward(g_rgb, g_alpha, None, skip_textures=True)[0], ref)
        t_oracle = time.time() - t0
        use_library('')
        fw = abi.forward_fused(faces, tex, S, 0.1, 100.0, eps, bg, 0, rgb, alpha, False)
        assert int((abi.host(fw['face_index_map']) != fn.face_index_map).sum()) == 0
        gr = abi.dev(g_rgb, torch.float32) if rgb else None
        ga = abi.dev(g_alpha, torch.float32) if alpha else None
        for tag in variants:
            lib = use_library(tag)
            # K6_FLAGS: the flag words to run (0 default kernel, 2 NR_FLAG_EXACT_GRADIENT, 128 NR_FLAG_K6_LEGACY, 8 NR_FLAG_K6_SCAN)
            for flags in [int(x) for x in os.environ.get('K6_FLAGS', '0 2').split()]:
                if (flags & 2) and tag:
                    continue  # the knobs do not touch the exact mode
                gf, us = k6_stage(lib, fw, gr, ga, flags, iters)
                ok = np.abs(ref) > 0
                err = np.abs(gf.astype(np.float64) - ref)
                print(json.dumps({
                    'scene': name, 'B': int(faces.shape[0]), 'F': int(faces.shape[1]), 'S': S, 'variant': tag or 'product',
                    'mode': 'exact' if flags & 2 else 'default', 'flags': flags, 'eps': eps, 'stage_us': us,
                    'err_floor_metric': H.rel_err(gf, ref), 'max_abs_err': float(err.max()), 'max_abs': float(np.abs(ref).max()),
                    'frac_within_1e-4_elementwise': float(np.mean(err[ok] <= 1e-4 * np.abs(ref[ok]))),
                    'frac_within_1e-5_elementwise': float(np.mean(err[ok] <= 1e-5 * np.abs(ref[ok]))),
                    'reference_float_sum_noise': noise, 'oracle_s': round(t_oracle, 1)}), flush=True)
        del fw, gr, ga
        torch.cuda.empty_cache()
    use_library('')


if __name__ == '__main__':
    main()
