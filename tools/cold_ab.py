"""GPU box: the cold-HBM rotation of bench.py (10 distinct buffer sets) for several builds of the library."""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from bench import make_boxes
dev = torch.device('cuda', 0)
n = 1_000_000
sets = [(make_boxes(n, 100 + 2 * k, dev), make_boxes(n, 101 + 2 * k, dev), torch.empty(n, device=dev)) for k in range(10)]
st = torch.cuda.current_stream().cuda_stream
for spec in sys.argv[1:]:
    label, path = spec.split('=')
    lib = ctypes.CDLL(os.path.join(ROOT, path))
    fn = lib.sph2pob_iou_aligned_f32
    fn.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int64] + [ctypes.c_int] * 5 + [ctypes.c_void_p]
    def run(reps, rot):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for r in range(reps):
            a, b, o = sets[r % 10 if rot else 0]
            fn(a.data_ptr(), b.data_ptr(), o.data_ptr(), n, 4, 0, 0, 0, 0, st)
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e3
    run(3000, False)
    warm = run(2000, False)
    run(500, True)
    cold = run(2000, True)
    print(f'{label}: same buffers {warm:.2f} us, 10 sets in rotation {cold:.2f} us')
