"""GPU box: capture combinations of the launchers into a hipGraph, each in its own process (a runtime crash in one
combination must not hide the others)."""
import faulthandler, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
COMBOS = ['iou', 'decode', 'loss', 'grad', 'iou+decode', 'iou+loss', 'iou+grad', 'decode+loss', 'iou+decode+loss', 'iou+decode+grad']
if len(sys.argv) == 1:
    for c in COMBOS:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), c], capture_output=True, text=True, timeout=120)
        print(c, 'rc', r.returncode, r.stdout.strip().splitlines()[-1:] , flush=True)
    sys.exit(0)
faulthandler.enable()
sys.path.insert(0, ROOT)
import torch
import sph_retina_amd as S
from oracle import oracle as O
n = 50000
b1 = torch.from_numpy(O.generate_boxes(n, 3)).cuda(); b2 = torch.from_numpy(O.generate_boxes(n, 4)).cuda()
anchors = b1.clone(); deltas = (torch.randn(n, 4, device='cuda') * 0.1).requires_grad_(True)
coder = S.DeltaXYWHSphBBoxCoder(target_stds=(0.1, 0.1, 0.2, 0.2)); loss_fn = S.Sph2PobIoULoss(mode='ciou')
parts = sys.argv[1].split('+')
def step():
    out = []
    if 'iou' in parts: out.append(S.sph2pob_standard_iou(b1, b2, is_aligned=True))
    if 'decode' in parts: out.append(coder.decode(anchors, deltas.detach()))
    if 'loss' in parts: out.append(loss_fn(b2, b1))
    if 'grad' in parts:
        loss = loss_fn(coder.decode(anchors, deltas), b1)
        out.append(torch.autograd.grad(loss, deltas)[0])
    return out
side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    step(); step()
torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
eager = [t.clone() for t in step()]
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    cap = step()
for t in cap: t.zero_()
g.replay(); torch.cuda.synchronize()
print('equal', all(torch.equal(a, b) for a, b in zip(eager, cap)), flush=True)
