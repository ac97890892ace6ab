"""Thin torch <-> C-ABI plumbing: device checks, pointers, current stream.  PyTorch is used for device
memory and streams only; all arithmetic happens in libsph2pob_hip.so."""
import ctypes

import torch

from . import _lib

import os

_VARIANT_CODES = {'standard': 0, 'efficient': 1, 'legacy': 2, 'sph_iou': 3, 'fov_iou': 4, 'unbiased': 5, 'naive': 6,
                  'naive_tan': 6 | 0x400}   # naive_tan: naive IoU with Sph2PlanarBoxTransform('sph2tan')
FLAG_REFERENCE_ORDER = 0x100
FLAG_NAIVE_TAN = 0x400   # naive variant: Sph2PlanarBoxTransform('sph2tan') instead of 'sph2pix'
_ARITHMETIC = os.environ.get('SPH2POB_ARITHMETIC', 'fast')
if _ARITHMETIC not in ('fast', 'robust', 'reference'):
    raise ValueError(f"SPH2POB_ARITHMETIC must be 'fast', 'robust' or 'reference', got {_ARITHMETIC!r}")
if _ARITHMETIC == 'robust':
    _ARITHMETIC = 'fast'


FLAG_ROBUST_PARALLEL = 0x200   # accepted by the ABI and ignored: the near-parallel safeguard has been always-on since round 2


def set_arithmetic(mode):
    """'fast' (default): the closed-form geometry core, near-parallel safeguard and NaN propagation included;
    'reference': the reference's fp32 operation order (3x the VALU work on the pairs that survive the cull; tracks the
    reference's rounding 3x more closely on nearby pairs); 'robust': round 1's name for the safeguard
    that is now part of 'fast' - kept as an alias."""
    global _ARITHMETIC
    assert mode in ('fast', 'robust', 'reference')
    _ARITHMETIC = 'fast' if mode == 'robust' else mode


def get_arithmetic():
    return _ARITHMETIC


class _Variants(dict):
    def __getitem__(self, k):
        return _VARIANT_CODES[k] | (FLAG_REFERENCE_ORDER if _ARITHMETIC == 'reference' else 0)


VARIANTS = _Variants(_VARIANT_CODES)
MODES = {'iou': 0, 'iof': 1}
EDGES = {'arc': 0, 'chord': 1, 'tangent': 2}
ANGLES = {'equator': 0, 'project': 1}


def require_hip(*tensors):
    """All operands on ONE device kind: MI355X tensors go to libsph2pob_hip.so, CPU tensors to the product's own host
    twins (libsph2pob_host.so, `*_cpu`: SURVEY §8b — the reference's operators run on CPU tensors too,
    tests/test_all_ious.py:88-104).  Anything else (mixed devices, another accelerator) raises."""
    kinds = {t.device.type for t in tensors}
    if len(kinds) > 1 or not kinds <= {'cuda', 'cpu'}:
        raise RuntimeError('sph_retina_amd: operands must all be MI355X (HIP) tensors or all be CPU tensors, got '
                           + ', '.join(sorted(str(t.device) for t in tensors)))


def as_f32(t):
    """Contiguous fp32 view/copy (never the caller's storage when a conversion is needed)."""
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


def as_f32_nograd(t):
    """as_f32 of a tensor the caller only reads: detached only when it is part of a graph (a detach() is a new tensor
    object, ~1 us of host time per operand of every call)."""
    return as_f32(t.detach() if t.requires_grad else t)


def ptr(t):
    """Device address as a plain integer (None = NULL): the launchers' argtypes are declared, so ctypes converts it
    without a c_void_p object being built for every argument of every call."""
    return t.data_ptr() if t is not None and t.numel() > 0 else None


_raw_stream = getattr(torch._C, '_cuda_getCurrentRawStream', None)


def stream_of(t):
    """hipStream_t of torch's current stream on t's device (the raw-handle query is ~20x cheaper than building a
    torch.cuda.Stream object on every launch); NULL for a CPU tensor (the host twins are synchronous)."""
    if t.device.type == 'cpu':
        return None
    if _raw_stream is not None:
        idx = t.device.index
        return ctypes.c_void_p(_raw_stream(torch.cuda.current_device() if idx is None else idx))
    return ctypes.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def raw_stream_of(device):
    """The same handle as a plain integer (what a ctypes `c_void_p` parameter accepts without building an object)."""
    if device.type == 'cpu':
        return None
    idx = device.index
    if idx is None:
        idx = torch.cuda.current_device()
    if _raw_stream is not None:
        return _raw_stream(idx)
    return torch.cuda.current_stream(device).cuda_stream


_FN_CACHE = {}


def call(name, device, *args):
    """Enqueue a launcher on the current stream of `device` and translate its return code."""
    if device.type == 'cpu':   # the product's host twin of the same entry point (synchronous)
        fn = _FN_CACHE.get((name, 'cpu'))
        if fn is None:
            if name not in _lib.HOST_TWINS:
                raise RuntimeError(f'{name} has no CPU twin in libsph2pob_host.so: this operator runs on MI355X tensors only')
            fn = _FN_CACHE[(name, 'cpu')] = getattr(_lib.host_lib(), name + '_cpu')
        rc = fn(*args)
        if rc:
            _lib.check(rc, name + '_cpu')
        return
    fn = _FN_CACHE.get(name)
    if fn is None:
        fn = _FN_CACHE[name] = getattr(_lib.lib(), name)
    if device.index is None or device.index == torch.cuda.current_device():
        rc = fn(*args)
    else:
        with torch.cuda.device(device):
            rc = fn(*args)
    if rc:
        _lib.check(rc, name)


_WORKSPACES = {}


def sum_workspace(device):
    """Scratch of the deterministic two-pass sum, one buffer per (device, stream): stream-ordered reuse is safe within a
    stream, a second stream gets its own.  A buffer first needed while the stream is CAPTURING is not cached: it would
    come from the graph's private pool and outlive it in this global table (round-1 VERDICT #6); the capture gets a plain
    temporary, which the pool keeps alive for the graph like any other tensor allocated inside the capture."""
    if device.type == 'cpu':
        return torch.empty((_lib.lib().sph2pob_sum_workspace_floats(),), dtype=torch.float32)
    idx = torch.cuda.current_device() if device.index is None else device.index
    key = (idx, _raw_stream(idx) if _raw_stream is not None else torch.cuda.current_stream(device).cuda_stream)
    ws = _WORKSPACES.get(key)
    if ws is None:
        ws = torch.empty((_lib.lib().sph2pob_sum_workspace_floats(),), dtype=torch.float32, device=device)
        if not torch.cuda.is_current_stream_capturing():
            _WORKSPACES[key] = ws
    return ws


_LOSS_WS = {}


def loss_sum_workspace(device, n):
    """Per-(device, stream) partial-sum scratch of `sph2pob_loss_fwd_sum_f32`, grown on demand (never shrunk); the same
    capture rule as sum_workspace."""
    if device.type == 'cpu':
        return torch.empty((8,), dtype=torch.float32)   # the host twins sum in place: the argument is not used
    idx = torch.cuda.current_device() if device.index is None else device.index
    key = (idx, raw_stream_of(device))
    need = (n + 255) // 256 + 1024
    ws = _LOSS_WS.get(key)
    if ws is None or ws.numel() < need:
        ws = torch.empty((max(need, 8192),), dtype=torch.float32, device=device)
        if not torch.cuda.is_current_stream_capturing():
            _LOSS_WS[key] = ws
    return ws


_ASSIGN_WS = {}


def assign_workspace(device, nbytes, state_bytes):
    """Per-(device, stream) buffers of the fused assigner -> (scratch, state).  `state` (per-GT accumulators + arrival counter)
    must be zero when a call is enqueued and every call leaves it zero, so it is zero-filled once when it is (re)allocated and
    then reused stream-ordered; both grow on demand and never shrink.  Same capture rule as sum_workspace."""
    idx = torch.cuda.current_device() if device.index is None else device.index
    key = (idx, raw_stream_of(device))
    words, swords = (nbytes + 7) // 8, (state_bytes + 7) // 8
    ws, state = _ASSIGN_WS.get(key, (None, None))
    if ws is None or ws.numel() < words:
        ws = torch.empty((max(words, 1 << 16),), dtype=torch.int64, device=device)
    if state is None or state.numel() < swords:
        state = torch.zeros((max(swords, 1 << 12),), dtype=torch.int64, device=device)
    if not torch.cuda.is_current_stream_capturing():
        _ASSIGN_WS[key] = (ws, state)
    return ws, state


def drop_assign_workspace(device):
    """After a failed call the accumulators may be dirty: forget the cached buffers of this device."""
    idx = torch.cuda.current_device() if device.index is None else device.index
    for key in [k for k in _ASSIGN_WS if k[0] == idx]:
        del _ASSIGN_WS[key]


_SCRATCH = {}


def scratch(device, nbytes):
    """Per-(device, stream) uint8 scratch that needs no initialisation (NMS workspace): stream-ordered reuse is safe within a
    stream; grown on demand, never shrunk.  Same capture rule as sum_workspace."""
    if device.type == 'cpu':
        return torch.empty((max(int(nbytes), 8),), dtype=torch.uint8)
    idx = torch.cuda.current_device() if device.index is None else device.index
    key = (idx, raw_stream_of(device))
    ws = _SCRATCH.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty((max(int(nbytes), 1 << 22),), dtype=torch.uint8, device=device)
        if not torch.cuda.is_current_stream_capturing():
            _SCRATCH[key] = ws
    return ws
