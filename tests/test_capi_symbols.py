"""CPU: the C-ABI library builds for gfx950, loads, and exports every symbol include/*.h declares (no compute)."""
import ctypes
import glob
import os
import re

import pytest

from conftest import ROOT


def declared_symbols():
    names = []
    for h in glob.glob(os.path.join(ROOT, 'include', '*.h')):
        src = open(h).read()
        src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
        names += re.findall(r'\b(sph2pob_[a-z0-9_]+)\s*\(', src)
    return sorted(set(names))


def test_library_builds_and_exports_every_declared_symbol():
    from sph_retina_amd import _lib
    _lib.build()
    handle = ctypes.CDLL(_lib.LIB_PATH)
    syms = declared_symbols()
    assert len(syms) >= 6
    for s in syms:
        assert hasattr(handle, s), f'{s} declared in include/ but not exported'
    assert sorted(_lib.SIGNATURES) == syms, 'python binding table and header disagree'
    lib = _lib.lib()
    assert lib.sph2pob_abi_version() == _lib.ABI_VERSION
    assert lib.sph2pob_target_arch() == b'gfx950'
    assert lib.sph2pob_error_string(-2).startswith(b'box_dim')


def test_argument_validation_without_gpu():
    """Launchers validate arguments before touching the device: error codes, n == 0 is a no-op."""
    from sph_retina_amd import _lib
    lib = _lib.lib()
    null = ctypes.c_void_p(0)
    assert lib.sph2pob_iou_aligned_f32(null, null, null, 0, 4, 0, 0, 0, 0, null) == 0
    assert lib.sph2pob_iou_aligned_f32(null, null, null, 10, 4, 0, 0, 0, 0, null) == -1
    assert lib.sph2pob_iou_aligned_f32(null, null, null, 10, 3, 0, 0, 0, 0, null) == -2
    assert lib.sph2pob_iou_aligned_f32(null, null, null, 10, 5, 2, 0, 0, 0, null) == -2  # legacy is BFoV only
    assert lib.sph2pob_iou_aligned_f32(null, null, null, 10, 4, 7, 0, 0, 0, null) == -3
    assert lib.sph2pob_iou_aligned_f32(null, null, null, -1, 4, 0, 0, 0, 0, null) == -4
    assert lib.sph2pob_iou_pairwise_f32(null, 0, null, 5, null, 4, 0, 0, 0, 0, null) == 0


def test_product_never_imports_oracle():
    """The product path must not route through the CPU oracle (or any CPU fallback)."""
    pkg = os.path.join(ROOT, 'sph_retina_amd')
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(('.py', '.hip', '.hpp', '.h', '.cpp')):
                text = open(os.path.join(dirpath, f)).read()
                assert 'import oracle' not in text and 'from oracle' not in text, os.path.join(dirpath, f)
                assert 'libsph2pob_oracle' not in text, os.path.join(dirpath, f)


def test_host_library_exports_every_cpu_twin_and_never_links_the_oracle():
    """libsph2pob_host.so (CPU twins `<name>_cpu`) is product code: built from sph_retina_amd/csrc only, it must
    export a twin for every name in _lib.HOST_TWINS with the HIP entry's signature, and must not include, link or load
    anything under oracle/."""
    import subprocess
    from sph_retina_amd import _lib
    _lib.build()
    handle = ctypes.CDLL(_lib.HOST_LIB_PATH)
    for name in _lib.HOST_TWINS:
        assert name in _lib.SIGNATURES and hasattr(handle, name + '_cpu'), name
    src = open(os.path.join(_lib.CSRC, _lib.HOST_SOURCES[0])).read()
    includes = re.findall(r'#include\s+"([^"]+)"', src)
    assert includes and all('oracle' not in i for i in includes), includes
    assert all(os.path.normpath(os.path.join(_lib.CSRC, i)).startswith((os.path.join(ROOT, 'sph_retina_amd'), os.path.join(ROOT, 'include')))
               for i in includes), includes
    needed = subprocess.run(['ldd', _lib.HOST_LIB_PATH], capture_output=True, text=True).stdout
    assert 'oracle' not in needed
    assert all('oracle' not in f for f in _lib.HOST_FLAGS + _lib.HOST_SOURCES)
    null = ctypes.c_void_p(0)
    lib = _lib.host_lib()
    assert lib.sph2pob_iou_aligned_f32_cpu(null, null, null, 0, 4, 0, 0, 0, 0, null) == 0
    assert lib.sph2pob_iou_aligned_f32_cpu(null, null, null, 10, 4, 0, 0, 0, 0, null) == -1
    assert lib.sph2pob_iou_aligned_f32_cpu(null, null, null, 10, 5, 2, 0, 0, 0, null) == -2
    assert lib.sph2pob_host_threads() >= 1


def test_mixed_devices_and_bad_options_fail_loudly():
    import torch
    import sph_retina_amd as S
    # empty inputs never reach a library (reference: sph_iou_api.py:56-57)
    assert S.sph2pob_standard_iou(torch.rand(0, 4), torch.rand(3, 4)).shape == (0, 3)
    assert S.sph2pob_efficient_iou(torch.rand(0, 4), torch.rand(0, 4), is_aligned=True).shape == (0, 1)
    with pytest.raises(AssertionError):
        S.sph2pob_standard_iou(torch.rand(4, 4), torch.rand(4, 4), mode='giou')
    with pytest.raises(RuntimeError, match='MI355X'):
        S.sph2pob_standard_iou(torch.rand(4, 4), torch.rand(4, 4, device='meta'))
    # operators without a CPU twin say so instead of falling back
    from sph_retina_amd.bbox.assigners import fused_assign
    with pytest.raises(RuntimeError, match='MI355X'):
        fused_assign(torch.rand(2, 4) * 50 + 20, torch.rand(9, 4) * 50 + 20)
