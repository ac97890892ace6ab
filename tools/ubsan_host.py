"""CPU: build the product's device math (tests/host_harness/harness.hip) for the host with UndefinedBehaviorSanitizer
(+ float-cast-overflow, bounds) and run it over adversarial inputs: uniform, identical, zero-size, polar, NaN / inf.
GPU sanitizers are not available on this pool; this is the CPU build the brief asks for.

    python tools/ubsan_host.py        (re-executes itself with the sanitizer runtime preloaded)
"""
import glob
import os
import subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, 'tests', 'host_harness', '_build', 'libhost_harness_ubsan.so')
if os.environ.get('SPH2POB_UBSAN_CHILD') != '1':
    os.makedirs(os.path.dirname(SO), exist_ok=True)
    subprocess.check_call(['hipcc', '--offload-arch=gfx950', '-O1', '-g', '-std=c++17', '-fPIC', '-shared', '-ffp-contract=off',
                           '-Xarch_host', '-fsanitize=undefined,float-cast-overflow,bounds', '-o', SO,
                           os.path.join(ROOT, 'tests', 'host_harness', 'harness.hip')], stderr=subprocess.DEVNULL)
    rt = glob.glob('/opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.ubsan_standalone-x86_64.so')[0]
    env = dict(os.environ, LD_PRELOAD=rt, UBSAN_OPTIONS='print_stacktrace=1:halt_on_error=0', SPH2POB_UBSAN_CHILD='1')
    r = subprocess.run([os.sys.executable, os.path.abspath(__file__)], env=env, cwd=ROOT, capture_output=True, text=True)
    print(r.stdout, end='')
    errs = [l for l in r.stderr.splitlines() if 'runtime error' in l]
    print('runtime errors reported:', len(errs))
    for l in errs[:20]:
        print(l)
    raise SystemExit(1 if errs or r.returncode else 0)
import ctypes, sys, numpy as np
sys.path.insert(0, '.')
from oracle import oracle as O
lib = ctypes.CDLL(SO)
def p(a): return a.ctypes.data_as(ctypes.c_void_p)
for dim in (4, 5):
    box = 'bfov' if dim == 4 else 'rbfov'
    sets = [('uniform', O.generate_boxes(5000, 0, box=box), O.generate_boxes(5000, 1, box=box))]
    b = O.generate_boxes(5000, 2, box=box)
    sets.append(('identical', b, b.copy()))
    z = b.copy(); z[:, 2:4] = 0; sets.append(('zero-size', b, z))
    pol = b.copy(); pol[:, 1] = 0; sets.append(('pole', pol, b))
    nan = b.copy(); nan[::7, 0] = np.nan; nan[::11, 3] = np.inf; sets.append(('nan/inf', nan, b))
    for name, b1, b2 in sets:
        b1 = np.ascontiguousarray(b1, np.float32); b2 = np.ascontiguousarray(b2, np.float32); n = len(b1)
        out = np.empty(n, np.float32)
        for v in (0, 1):
            lib.harness_iou_fast(p(b1), p(b2), ctypes.c_int64(n), dim, v, 0, 0, p(out))
            lib.harness_iou(p(b1), p(b2), ctypes.c_int64(n), dim, v, 0, 0, 0, p(out))
        for w in (0, 1, 2):
            lib.harness_extra_iou(p(b1), p(b2), ctypes.c_int64(n), dim, w, p(out))
        loss, iou = np.empty(n, np.float32), np.empty(n, np.float32); gp, gt = np.empty((n, dim), np.float32), np.empty((n, dim), np.float32)
        for m in range(4):
            for f in (0, 1):
                lib.harness_loss(p(b1), p(b2), ctypes.c_int64(n), dim, m, ctypes.c_float(1e-6), p(loss), p(iou), p(gp), p(gt), f)
        # adjoints of the transforms: closed form (standard / efficient) and forward-mode (legacy, project; +- jitter)
        g1, g2 = np.ones((n, 5), np.float32), np.full((n, 5), -0.5, np.float32)
        o1, o2 = np.empty((n, dim), np.float32), np.empty((n, dim), np.float32)
        for v in (0, 1):
            for jit in (0, 1):
                lib.harness_transform_bwd(p(b1), p(b2), p(g1), p(g2), ctypes.c_int64(n), dim, v, 0, jit, p(o1), p(o2))
                lib.harness_transform_bwd_general(p(b1), p(b2), p(g1), p(g2), ctypes.c_int64(n), dim, v, 1, 1, jit, p(o1), p(o2))
        if dim == 4:
            for jit in (0, 1):
                lib.harness_transform_bwd_general(p(b1), p(b2), p(g1), p(g2), ctypes.c_int64(n), 4, 2, 0, 0, jit, p(o1), p(o2))
        print(dim, name, 'ok', flush=True)
