"""GPU box: where the host time of the Python boundary goes (cProfile over the autograd loss step and the IoU call)."""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sph_retina_amd as S
from sph_retina_amd.losses import Sph2PobIoULoss
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
g = torch.Generator().manual_seed(0)
t = torch.rand((n, 5), generator=g).cuda() * torch.tensor([360., 180., 80., 80., 100.]).cuda() + torch.tensor([0., 0., 5., 5., -50.]).cuda()
p = (t + torch.randn_like(t)).requires_grad_(True)
loss = Sph2PobIoULoss(mode='ciou')
def step():
    p.grad = None
    loss(p, t).backward()
def iou():
    S.sph2pob_standard_iou(p.detach()[:, :4], t[:, :4], is_aligned=True)
for fn in (step, iou):
    for _ in range(200): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(2000): fn()
    torch.cuda.synchronize()
    print(fn.__name__, 'host+device us per call: %.1f' % ((time.perf_counter() - t0) / 2000 * 1e6))
    pr = cProfile.Profile(); pr.enable()
    for _ in range(2000): fn()
    pr.disable(); torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats('tottime').print_stats(14)
