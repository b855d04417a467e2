"""Build libnr_hip.so (the C-ABI library of include/nr_hip.h) for gfx950 with hipcc, in-tree.

hipcc cross-compiles without a GPU, so this runs in the authoring container; the resulting .so is
git-ignored but travels to the GPU box with the working tree.
"""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SOURCES = [os.path.join(HERE, 'csrc', n) for n in ('nr_forward.hip', 'nr_backward_pixel_map.hip', 'nr_backward_gather.hip', 'nr_geometry.hip',
                                                      'nr_image.hip', 'nr_frontend.hip', 'nr_texture_io.hip', 'nr_optim.hip')]
HEADERS = [os.path.join(os.path.dirname(HERE), 'include', 'nr_hip.h'), os.path.join(os.path.dirname(HERE), 'include', 'nr_hip_profile.h'),
           os.path.join(HERE, 'csrc', 'nr_device.h'),
           os.path.join(HERE, 'csrc', 'nr_k6_tune.h'), os.path.join(HERE, 'csrc', 'nr_band_lines.h')]
LIB_PATH = os.path.join(HERE, 'libnr_hip.so')
# the measurement build: the same sources with -DNR_PROFILE_HOOK (include/nr_hip_profile.h); bench.py times the dominant kernel with it
PROFILE_LIB_PATH = os.path.join(HERE, 'libnr_hip_prof.so')

# -ffp-contract=off + correctly rounded division: the parity contract (DESIGN.md "Numerics").
# -munsafe-fp-atomics: hardware global_atomic_add_f32 instead of a CAS loop (torch memory is coarse-grained).
HIPCC_FLAGS = [
    '--offload-arch=gfx950', '-O3', '-std=c++17', '-ffp-contract=off',
    '-fhip-fp32-correctly-rounded-divide-sqrt', '-munsafe-fp-atomics', '-fno-fast-math',
    # no SLP vectorisation: it turns the unrolled per-pixel arithmetic of the K6 sweeps into v_pk_* operations glued
    # together with register moves -- packed FP32 issues at half rate on this part, so that only costs
    '-fno-slp-vectorize',
    '-fPIC', '-shared', '-fvisibility=hidden',
]


def hipcc():
    for c in (os.environ.get('HIPCC'), '/opt/rocm/bin/hipcc', 'hipcc'):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return 'hipcc'


def needs_build():
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    return any(os.path.getmtime(p) > t for p in SOURCES + HEADERS + [os.path.abspath(__file__)])


def build(force=False, verbose=False):
    """Compile if the library is missing or older than its sources. Returns the library path."""
    if force or needs_build():
        cmd = [hipcc()] + HIPCC_FLAGS + SOURCES + ['-o', LIB_PATH]
        if verbose:
            print(' '.join(cmd))
        subprocess.check_call(cmd)
    return LIB_PATH


def build_profile(force=False, verbose=False):
    """libnr_hip_prof.so: the product's sources plus the band-kernel timing hook (include/nr_hip_profile.h).  Loaded only by
    bench.py's roofline measurement and by the hook's test (neural_renderer_amd._lib.load_profile)."""
    if force or not os.path.exists(PROFILE_LIB_PATH) or \
            any(os.path.getmtime(p) > os.path.getmtime(PROFILE_LIB_PATH) for p in SOURCES + HEADERS + [os.path.abspath(__file__)]):
        cmd = [hipcc()] + HIPCC_FLAGS + ['-DNR_PROFILE_HOOK=1'] + SOURCES + ['-o', PROFILE_LIB_PATH]
        if verbose:
            print(' '.join(cmd))
        subprocess.check_call(cmd)
    return PROFILE_LIB_PATH


def build_variant(tag, defines, verbose=False, sources=None, flags=None):
    """Development aid: the same sources with extra -D macros into libnr_hip_<tag>.so, so that one GPU session can time
    alternatives of a kernel side by side (scripts pick the library through the NR_HIP_LIB environment variable, read by
    neural_renderer_amd._lib -- the library itself reads no environment).  Not used by the product."""
    out = os.path.join(HERE, 'libnr_hip_%s.so' % tag)
    cmd = [hipcc()] + (flags or HIPCC_FLAGS) + ['-D%s' % d for d in defines] + (sources or SOURCES) + ['-o', out]
    if verbose:
        print(' '.join(cmd))
    subprocess.check_call(cmd)
    return out


if __name__ == '__main__':
    import sys
    if len(sys.argv) > 2:  # python -m neural_renderer_amd._build <tag> NAME=VALUE ...
        print(build_variant(sys.argv[1], sys.argv[2:], verbose=True))
    else:
        print(build(force=True, verbose=True))
        print(build_profile(force=True, verbose=True))
