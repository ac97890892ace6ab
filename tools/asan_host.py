#!/usr/bin/env python3
"""CPU: the host twins (libsph2pob_host.so) built with AddressSanitizer + UndefinedBehaviorSanitizer and driven by the CPU tier's
own tests of the CPU tensors (tests/test_cpu_twins.py, the coder / L1 / unbiased / naive CPU tests) plus ragged and empty sizes.
GPU sanitizers are not available on this pool; the twins are the kernels' own host-compiled arithmetic plus host loops (thread
chunks, NMS segments, assigner epilogue) that exist only here.

    python tools/asan_host.py        (builds build/asan/libsph2pob_host.so, re-runs itself with the sanitizer runtime preloaded)
"""
import glob
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, 'build', 'asan', 'libsph2pob_host.so')
sys.path.insert(0, ROOT)

if os.environ.get('SPH2POB_ASAN_CHILD') != '1':
    from sph_retina_amd import _lib
    os.makedirs(os.path.dirname(SO), exist_ok=True)
    flags = [f for f in _lib.HOST_FLAGS if f != '-O2'] + ['-O1', '-g', '-fno-omit-frame-pointer',
                                                          '-fsanitize=address,undefined,float-cast-overflow,bounds', '-shared-libsan']
    cmd = ['hipcc'] + flags + ['-o', SO] + [os.path.join(_lib.CSRC, s) for s in _lib.HOST_SOURCES]
    subprocess.check_call(cmd, cwd=_lib.CSRC, stderr=subprocess.DEVNULL)
    rt = glob.glob('/opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so')[0]
    env = dict(os.environ, LD_PRELOAD=rt, ASAN_OPTIONS='detect_leaks=0:halt_on_error=0:abort_on_error=0',
               UBSAN_OPTIONS='print_stacktrace=1:halt_on_error=0', SPH2POB_ASAN_CHILD='1')
    r = subprocess.run([sys.executable, os.path.abspath(__file__)], env=env, cwd=ROOT, capture_output=True, text=True)
    print(r.stdout[-3000:], end='')
    errs = [l for l in r.stderr.splitlines() if 'runtime error' in l or 'AddressSanitizer' in l]
    print('sanitizer reports:', len(errs))
    for l in errs[:30]:
        print(l)
    raise SystemExit(1 if errs or r.returncode else 0)

# ---- child: the sanitizer runtime is loaded; point the package at the instrumented twins and run the CPU tests of CPU tensors ----
from sph_retina_amd import _lib   # noqa: E402
_lib.HOST_LIB_PATH = SO
import numpy as np      # noqa: E402
import torch            # noqa: E402
import pytest           # noqa: E402
import sph_retina_amd as S   # noqa: E402
assert _lib.host_lib()._name == SO
# ragged / empty / single-row sizes through every operator family (thread chunking, segment ends)
g = torch.Generator().manual_seed(1)
for n in (0, 1, 2, 63, 64, 65, 2047, 2049, 4097):
    for dim in (4, 5):
        u = torch.rand((n, 5), generator=g)
        b1 = torch.stack([u[:, 0] * 360, u[:, 1] * 180, 1 + u[:, 2] * 99, 1 + u[:, 3] * 99, -90 + u[:, 4] * 180], 1)[:, :dim].contiguous()
        b2 = (b1 + torch.randn((n, dim), generator=g) * 3).contiguous()
        b2[:, 1:4] = b2[:, 1:4].clamp(1, 179)
        S.sph2pob_standard_iou(b1, b2, is_aligned=True)
        S.sph2pob_efficient_iou(b1[:7], b2)
        if dim == 4:
            S.sph2pob_legacy_iou(b1, b2, is_aligned=True)
            S.SphOverlaps2D(backend='unbiased_iou')(b1[:5], b2[:9])
        p = b1.clone().requires_grad_(True)
        loss = S.Sph2PobIoULoss(mode='ciou')(p, b2)
        if n:
            loss.backward()
        if n:
            from sph_retina_amd.bbox.nms import SphNMS
            from sph_retina_amd.bbox.assigners import SphMaxIoUAssigner
            SphNMS()(b1, torch.rand(n, generator=g), torch.randint(0, 3, (n,), generator=g), dict(iou_threshold=0.5, max_num=10))
            if dim == 4:
                SphMaxIoUAssigner(pos_iou_thr=0.5, neg_iou_thr=0.4, min_pos_iou=0.0).assign(b2, b1[:min(n, 9)], gt_labels=torch.arange(min(n, 9)))
print('ragged sizes done', flush=True)
rc = pytest.main(['-x', '-q', '-m', 'not gpu', '-p', 'no:cacheprovider', 'tests/test_cpu_twins.py', 'tests/test_coder.py', 'tests/test_l1_loss.py',
                  'tests/test_unbiased_naive.py'])
raise SystemExit(int(rc))
