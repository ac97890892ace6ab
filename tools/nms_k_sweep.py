"""GPU box (under rocprofv3 --kernel-trace): the host-free batched NMS at several candidate counts; tools/trace_summary.py then
gives the four kernels per grid size (profiles/r05t_trace_nms_k_sweep.txt)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sph_retina_amd import _lib, _torch_glue as G
lib = _lib.lib()
st = G.raw_stream_of(torch.device('cuda', 0))
for k in (500, 2000, 3000, 5000, 8000, 12000, 16000):
    g = torch.Generator().manual_seed(k)
    b = torch.stack([torch.rand(k, generator=g) * 360, torch.rand(k, generator=g) * 140 + 20, torch.rand(k, generator=g) * 20 + 2, torch.rand(k, generator=g) * 20 + 2], 1).cuda()
    s = torch.rand(k, generator=g).cuda()
    c = torch.randint(0, 37, (k,), generator=g).cuda()
    ws = torch.empty(lib.sph2pob_batched_nms_workspace_bytes(k, 4), dtype=torch.uint8, device='cuda')
    ko, do, stt = torch.zeros(100, dtype=torch.int64, device='cuda'), torch.zeros((100, 5), device='cuda'), torch.zeros(1, dtype=torch.int32, device='cuda')
    for _ in range(300):
        lib.sph2pob_batched_nms_f32(G.ptr(b), G.ptr(s), G.ptr(c), k, 4, 1, 0.5, 100, G.ptr(ws), G.ptr(ko), G.ptr(do), G.ptr(stt), st)
    torch.cuda.synchronize()
