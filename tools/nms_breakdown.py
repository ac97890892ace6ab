"""GPU box: where the time of a 5000-box single-class SphNMS call goes (kernels through the C ABI vs host-side torch)."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import sph_retina_amd as S
from sph_retina_amd import _lib, _torch_glue as G
from sph_retina_amd.bbox.nms import SphNMS
k = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
rng = np.random.default_rng(4)
c = np.stack([rng.random(300) * 360, rng.random(300) * 180, 5 + rng.random(300) * 55, 5 + rng.random(300) * 55], 1).astype(np.float32)
b = c[rng.integers(0, 300, k)] + rng.standard_normal((k, 4)).astype(np.float32) * 2
b[:, 0] %= 360; b[:, 1] = b[:, 1].clip(1, 179); b[:, 2:] = b[:, 2:].clip(2, 120)
boxes, scores, idxs = torch.from_numpy(b).cuda(), torch.rand(k).cuda(), torch.zeros(k, dtype=torch.long).cuda()
nms = SphNMS()
def t(fn, reps=30):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e6
print('SphNMS call            %.1f us' % t(lambda: nms(boxes, scores, idxs, dict(iou_threshold=0.5, max_num=100))))
order = torch.argsort(scores, descending=True, stable=True)
sb = boxes[order].contiguous(); lib = _lib.lib()
keep = torch.empty(k, dtype=torch.uint8, device='cuda')
ws = torch.empty(lib.sph2pob_nms_workspace_bytes(k) // 8, dtype=torch.int64, device='cuda')
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
print('mask + sweep (C ABI)   %.1f us' % t(lambda: lib.sph2pob_nms_f32(G.ptr(sb), None, ctypes.c_int64(k), 4, 1, ctypes.c_float(0.5), G.ptr(ws), G.ptr(keep), st)))
print('stable argsort         %.1f us' % t(lambda: torch.argsort(scores, descending=True, stable=True)))
print('gather boxes[order]    %.1f us' % t(lambda: boxes[order]))
print('nonzero (sync)         %.1f us' % t(lambda: keep.bool().nonzero()))
