"""ctypes binding of libsph2pob_hip.so (the C ABI declared in include/sph2pob_hip.h).

No torch extension and no silent fallback: if a shared library is missing or a symbol is absent the product fails
loudly.  ``build()`` cross-compiles libsph2pob_hip.so for gfx950 with hipcc (works without a GPU) and compiles
libsph2pob_host.so — the CPU twins (`*_cpu`) of the same entry points, the product's own __host__ __device__ arithmetic
instantiated for the host (csrc/sph2pob_host.hip) — with the same compiler; both live in-tree under
sph_retina_amd/lib/ so they travel with the source tree.  CPU tensors are served by the host library, device tensors by
the HIP one; neither ever touches oracle/.
"""
import ctypes
import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, 'csrc')
LIB_DIR = os.path.join(_HERE, 'lib')
LIB_PATH = os.path.join(LIB_DIR, 'libsph2pob_hip.so')
HOST_LIB_PATH = os.path.join(LIB_DIR, 'libsph2pob_host.so')
SOURCES = ['sph2pob_iou.hip', 'sph2pob_assign.hip', 'sph2pob_loss.hip', 'sph2pob_nms.hip', 'sph2pob_coder.hip']
HOST_SOURCES = ['sph2pob_host.hip']
HOST_FLAGS = ['--offload-arch=gfx950', '--cuda-host-only', '-O2', '-std=c++17', '-fPIC', '-shared', '-ffp-contract=off', '-pthread']


def _host_has_fma():
    try:
        with open('/proc/cpuinfo') as f:
            return any(line.startswith('flags') and ' fma ' in line + ' ' for line in f)
    except OSError:
        return False


# the twins' explicit fmaf() calls (the kernels' v_fma_f32) as the hardware instruction instead of a libm call per product:
# the same bits (both round once), 1.6x the pairs per second per core.  Every MI355X host has it; a build machine without it
# keeps the libm call.
if _host_has_fma():
    HOST_FLAGS.append('-mfma')
# the entry points that have a CPU twin `<name>_cpu` with the same signature (the stream argument is ignored)
HOST_TWINS = ['sph2pob_iou_aligned_f32', 'sph2pob_iou_pairwise_f32', 'sph2pob_planar_iou_f32', 'sph2pob_transform_f32',
              'sph2pob_transform_bwd_f32', 'sph2pob_transform_bwd_general_f32', 'sph2pob_loss_fwd_f32', 'sph2pob_loss_bwd_f32',
              'sph2pob_loss_fwd_sum_f32', 'sph2pob_loss_fwd_grad_f32', 'sph2pob_loss_grad_scale_f32', 'sph2pob_sum_f32',
              'sph2pob_nms_segmented_f32', 'sph2pob_nms_f32', 'sph2pob_assign_f32', 'sph2pob_coder_encode_f32',
              'sph2pob_coder_decode_f32', 'sph2pob_coder_decode_bwd_f32', 'sph2pob_obb_l1_fwd_f32', 'sph2pob_obb_l1_bwd_f32']
HEADERS = ['sph2pob_device.hpp', 'sph2pob_loss.hpp', 'sph2pob_fast.hpp', 'sph2pob_unbiased.hpp', 'sph2pob_coder.hpp', 'sph2pob_kernels_common.hpp', os.path.join('..', '..', 'include', 'sph2pob_hip.h')]
# -fno-slp-vectorize: hipcc otherwise pairs scalar fp32 mul/add into v_pk_* (+ v_mov shuffles); packed fp32 issues at
# half the rate of plain VALU on gfx950 (tools/ubench/valu_rate2.hip), measured 12 % slower on the dominant kernel
# -amdgpu-kernarg-preload-count: gfx950 hands the first kernel arguments to a wave in SGPRs at launch instead of making
# every wave fetch them with s_load (two dependent scalar loads stood in front of the first global load of the dominant
# kernel): 8.24 -> 8.04 us per 1 M pairs, 4.29 -> 4.11 at 250 k (profiles/r02w_ab_*.log); the compiler keeps a
# compatible entry for firmware without the feature
HIPCC_FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-shared', '-ffp-contract=off',
               '-fno-slp-vectorize', '-mllvm', '-amdgpu-kernarg-preload-count=16'] + \
    os.environ.get('SPH2POB_EXTRA_HIPCC_FLAGS', '').split()

_c_f32p = ctypes.c_void_p
_i64 = ctypes.c_int64
_int = ctypes.c_int

# name -> argtypes (restype is always int unless listed in _RESTYPES); mirrors include/sph2pob_hip.h
SIGNATURES = {
    'sph2pob_abi_version': [],
    'sph2pob_target_arch': [],
    'sph2pob_error_string': [_int],
    'sph2pob_iou_aligned_f32': [_c_f32p, _c_f32p, _c_f32p, _i64, _int, _int, _int, _int, _int, ctypes.c_void_p],
    'sph2pob_iou_pairwise_f32': [_c_f32p, _i64, _c_f32p, _i64, _c_f32p, _int, _int, _int, _int, _int,
                                 ctypes.c_void_p],
    'sph2pob_planar_iou_f32': [_c_f32p, _i64, _c_f32p, _i64, _c_f32p, _int, _int, ctypes.c_void_p],
    'sph2pob_transform_f32': [_c_f32p, _c_f32p, _c_f32p, _c_f32p, _i64, _int, _int, _int, _int, _int,
                              ctypes.c_void_p],
    'sph2pob_transform_bwd_f32': [_c_f32p] * 6 + [_i64, _int, _int, _int, _int, ctypes.c_void_p],
    'sph2pob_transform_bwd_general_f32': [_c_f32p] * 6 + [_i64, _int, _int, _int, _int, _int, ctypes.c_void_p],
    'sph2pob_loss_fwd_f32': [_c_f32p, _c_f32p, _c_f32p, _int, ctypes.c_float, _c_f32p, _c_f32p, _i64, _int, _int,
                             ctypes.c_float,
                             ctypes.c_void_p],
    'sph2pob_loss_bwd_f32': [_c_f32p, _c_f32p, _c_f32p, _int, _c_f32p, _int, ctypes.c_float, _c_f32p, _c_f32p, _i64,
                             _int, _int, ctypes.c_float, ctypes.c_void_p],
    'sph2pob_loss_sum_workspace_floats': [_i64],
    'sph2pob_loss_fwd_sum_f32': [_c_f32p, _c_f32p, _c_f32p, _int, ctypes.c_float, _c_f32p, _c_f32p, _i64, _int, _int,
                                 ctypes.c_float, ctypes.c_void_p],
    'sph2pob_loss_fwd_grad_f32': [_c_f32p, _c_f32p, _c_f32p, _int, ctypes.c_float, _c_f32p, _c_f32p, _c_f32p, _c_f32p, _c_f32p,
                                  _i64, _int, _int, ctypes.c_float, ctypes.c_void_p],
    'sph2pob_loss_grad_scale_f32': [_c_f32p, _c_f32p, _int, _c_f32p, _i64, _int, ctypes.c_void_p],
    'sph2pob_sum_workspace_floats': [],
    'sph2pob_sum_f32': [_c_f32p, _i64, ctypes.c_float, _c_f32p, _c_f32p, ctypes.c_void_p],
    'sph2pob_assign_workspace_bytes': [_i64, _i64],
    'sph2pob_assign_f32': [_c_f32p, _i64, _i64, ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_float, _int, _int,
                           ctypes.c_void_p] + [ctypes.c_void_p] * 7 + [ctypes.c_void_p],
    'sph2pob_iou_assign_workspace_bytes': [_i64, _i64],
    'sph2pob_iou_assign_reduce_f32': [_c_f32p, _i64, _c_f32p, _i64, _int, _int, _int, ctypes.c_void_p, _i64, _c_f32p,
                                      ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p],
    'sph2pob_iou_assign_finalize_f32': [_c_f32p, _i64, _c_f32p, _i64, _int, _int, _int, _i64, ctypes.c_void_p] +
                                       [ctypes.c_float] * 4 + [_int, _int] + [ctypes.c_void_p] * 9,
    'sph2pob_iou_assign_f32': [_c_f32p, _i64, _c_f32p, _i64, _int, _int, _int, ctypes.c_void_p, _c_f32p] +
                              [ctypes.c_float] * 4 + [_int, _int] + [ctypes.c_void_p] * 10,
    'sph2pob_iou_assign_state_bytes': [_i64, _i64],
    'sph2pob_nms_max_boxes': [],
    'sph2pob_nms_workspace_bytes': [_i64],
    'sph2pob_nms_f32': [_c_f32p, ctypes.c_void_p, _i64, _int, _int, ctypes.c_float, ctypes.c_void_p, ctypes.c_void_p,
                        ctypes.c_void_p],
    'sph2pob_nms_segmented_workspace_bytes': [_i64, _i64],
    'sph2pob_nms_segmented_f32': [_c_f32p, ctypes.c_void_p, _i64, _int, _int, ctypes.c_float, _i64, ctypes.c_void_p,
                                  ctypes.c_void_p, ctypes.c_void_p],
    'sph2pob_batched_nms_max_boxes': [],
    'sph2pob_batched_nms_workspace_bytes': [_i64, _int],
    'sph2pob_batched_nms_f32': [_c_f32p, _c_f32p, ctypes.c_void_p, _i64, _int, _int, ctypes.c_float, _i64] + [ctypes.c_void_p] * 5,
    'sph2pob_coder_encode_f32': [_c_f32p, _c_f32p, ctypes.c_void_p, ctypes.c_void_p, _c_f32p, _i64, _int,
                                 ctypes.c_void_p],
    'sph2pob_coder_decode_f32': [_c_f32p, _c_f32p, ctypes.c_void_p, ctypes.c_void_p, _c_f32p, _i64, _int, _int,
                                 ctypes.c_float, _int, ctypes.c_float, ctypes.c_void_p],
    'sph2pob_coder_decode_bwd_f32': [_c_f32p, _c_f32p, _c_f32p, ctypes.c_void_p, ctypes.c_void_p, _c_f32p, _i64, _int,
                                     _int, ctypes.c_float, _int, ctypes.c_float, ctypes.c_void_p],
    'sph2pob_obb_l1_fwd_f32': [_c_f32p, _c_f32p, _c_f32p, ctypes.c_float, _c_f32p, _i64, _int, ctypes.c_void_p],
    'sph2pob_obb_l1_bwd_f32': [_c_f32p, _c_f32p, _c_f32p, _c_f32p, ctypes.c_float, _c_f32p, _c_f32p, _i64, _int,
                               ctypes.c_void_p],
}
_RESTYPES = {'sph2pob_loss_sum_workspace_floats': ctypes.c_int64, 'sph2pob_target_arch': ctypes.c_char_p, 'sph2pob_error_string': ctypes.c_char_p,
             'sph2pob_nms_workspace_bytes': ctypes.c_int64, 'sph2pob_nms_segmented_workspace_bytes': ctypes.c_int64, 'sph2pob_assign_workspace_bytes': ctypes.c_int64,
             'sph2pob_iou_assign_workspace_bytes': ctypes.c_int64, 'sph2pob_iou_assign_state_bytes': ctypes.c_int64,
             'sph2pob_batched_nms_workspace_bytes': ctypes.c_int64}

ABI_VERSION = 1


class Sph2PobLibraryError(RuntimeError):
    pass


def _older_than(path, sources):
    if not os.path.exists(path):
        return True
    t = os.path.getmtime(path)
    deps = [os.path.join(CSRC, s) for s in sources] + [os.path.normpath(os.path.join(CSRC, h)) for h in HEADERS]
    return any(os.path.exists(d) and os.path.getmtime(d) > t for d in deps)


def _stale():
    return _older_than(LIB_PATH, SOURCES) or _older_than(HOST_LIB_PATH, HOST_SOURCES)


def _compile(flags, sources, out, verbose):
    """One hipcc -c per translation unit, in parallel, then one link.  Objects are kept under lib/obj/ and reused while they
    are newer than their source and every header and were built with the same flags (an edit to one .hip recompiles one unit)."""
    import hashlib
    from concurrent.futures import ThreadPoolExecutor
    hipcc = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    if not os.path.exists(hipcc):
        raise Sph2PobLibraryError(f'hipcc not found: cannot build {os.path.basename(out)}')
    os.makedirs(LIB_DIR, exist_ok=True)
    obj_dir = os.path.join(LIB_DIR, 'obj')
    os.makedirs(obj_dir, exist_ok=True)
    cflags = [f for f in flags if f != '-shared']
    tag = hashlib.sha1(' '.join(cflags).encode()).hexdigest()[:10]
    deps = [os.path.join(CSRC, h) for h in HEADERS]
    newest_header = max(os.path.getmtime(d) for d in deps if os.path.exists(d))

    def unit(src):
        path = os.path.join(CSRC, src)
        obj = os.path.join(obj_dir, f'{os.path.splitext(src)[0]}.{tag}.o')
        if os.path.exists(obj) and os.path.getmtime(obj) > max(os.path.getmtime(path), newest_header):
            return obj
        tmp = f'{obj}.tmp.{os.getpid()}'
        cmd = [hipcc] + cflags + ['-c', '-o', tmp, path]
        if verbose:
            print(' '.join(cmd))
        try:
            subprocess.check_call(cmd, cwd=CSRC)
            os.replace(tmp, obj)
        finally:
            if os.path.exists(tmp):
                os.remove(tmp)
        return obj

    with ThreadPoolExecutor(max_workers=min(len(sources), max(1, min(os.cpu_count() or 1, 8)))) as pool:
        objs = list(pool.map(unit, sources))
    tmp = f'{out}.tmp.{os.getpid()}'   # linked aside and renamed: other ranks / processes never see a partial file
    link = [hipcc] + [f for f in flags if f.startswith('--offload-arch') or f in ('-shared', '-fPIC', '-pthread', '--cuda-host-only')] + ['-o', tmp] + objs
    if verbose:
        print(' '.join(link))
    try:
        subprocess.check_call(link, cwd=CSRC)
        os.replace(tmp, out)
    finally:
        if os.path.exists(tmp):
            os.remove(tmp)


def build(force=False, verbose=False):
    """hipcc --offload-arch=gfx950 ... -> sph_retina_amd/lib/libsph2pob_hip.so (no GPU needed) and the host twins
    -> sph_retina_amd/lib/libsph2pob_host.so."""
    if force or _older_than(LIB_PATH, SOURCES):
        _compile(HIPCC_FLAGS, SOURCES, LIB_PATH, verbose)
    if force or _older_than(HOST_LIB_PATH, HOST_SOURCES):
        _compile(HOST_FLAGS, HOST_SOURCES, HOST_LIB_PATH, verbose)
    return LIB_PATH


_LIB = None


def lib():
    """The loaded library with typed entry points; raises Sph2PobLibraryError when it cannot be used."""
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(LIB_PATH):
        raise Sph2PobLibraryError(
            f'{LIB_PATH} is missing: build it with `python -c "import __graft_entry__ as g; g.build()"` '
            '(hipcc --offload-arch=gfx950).  There is no CPU / eager fallback for the Sph2Pob path.')
    try:
        handle = ctypes.CDLL(LIB_PATH)
    except OSError as e:
        raise Sph2PobLibraryError(f'cannot load {LIB_PATH}: {e}') from e
    for name, argtypes in SIGNATURES.items():
        try:
            fn = getattr(handle, name)
        except AttributeError as e:
            raise Sph2PobLibraryError(f'{LIB_PATH} does not export {name}') from e
        fn.argtypes = argtypes
        fn.restype = _RESTYPES.get(name, _int)
    if handle.sph2pob_abi_version() != ABI_VERSION:
        raise Sph2PobLibraryError('libsph2pob_hip.so ABI version mismatch: rebuild the library')
    _LIB = handle
    return _LIB


_HOST = None


def host_lib():
    """libsph2pob_host.so with typed `<name>_cpu` entry points (the CPU twins); raises Sph2PobLibraryError when unusable."""
    global _HOST
    if _HOST is not None:
        return _HOST
    if not os.path.exists(HOST_LIB_PATH):
        raise Sph2PobLibraryError(
            f'{HOST_LIB_PATH} is missing: build it with `python -c "import __graft_entry__ as g; g.build()"`.  CPU tensors '
            'are served by this library only (there is no eager / oracle fallback).')
    try:
        handle = ctypes.CDLL(HOST_LIB_PATH)
    except OSError as e:
        raise Sph2PobLibraryError(f'cannot load {HOST_LIB_PATH}: {e}') from e
    for name in HOST_TWINS:
        try:
            fn = getattr(handle, name + '_cpu')
        except AttributeError as e:
            raise Sph2PobLibraryError(f'{HOST_LIB_PATH} does not export {name}_cpu') from e
        fn.argtypes = SIGNATURES[name]
        fn.restype = _int
    handle.sph2pob_host_threads.restype = _int
    _HOST = handle
    return _HOST


def check(rc, what):
    if rc != 0:
        msg = lib().sph2pob_error_string(int(rc))
        raise Sph2PobLibraryError(f'{what} failed: {msg.decode() if msg else rc} (code {rc})')
