import os, sys
import torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import sph_retina_amd as S
from sph_retina_amd.losses import Sph2PobIoULoss
b1 = torch.tensor([[1., 49.131344, 88.38273, 6.3124614, -83.81888]], device='cuda'); b2 = torch.tensor([[0., 49.994923, 87.12384, 8.817537, -83.099594]], device='cuda')
torch.set_printoptions(precision=8)
for arith in ('fast', 'reference'):
    S.set_arithmetic(arith)
    print(arith, 'iou', S.sph2pob_standard_iou(b1, b2, is_aligned=True).item(), 'eff', S.sph2pob_efficient_iou(b1, b2, is_aligned=True).item())
    p = b1.clone().requires_grad_(True)
    l = Sph2PobIoULoss(mode='iou', reduction='none')(p, b2); l.sum().backward()
    print('   loss', l.item(), 'grad', p.grad)
    P, T = S.iou.sph_iou_api._transform('standard', b1, b2, 'rad', 'arc', 'equator', jitter=True)
    print('   planar', P, T, 'da', (P[0,4]-T[0,4]).item(), 'sin/cos', torch.sin(P[0,4]).item(), torch.cos(P[0,4]).item(), torch.sin(T[0,4]).item(), torch.cos(T[0,4]).item())
