import sys, ctypes; import os; R=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,R); sys.path.insert(0,R+'/tests')
import numpy as np, torch
from conftest import load_golden
import sph_retina_amd as S
from sph_retina_amd import _lib, _torch_glue as G
g = load_golden('nms')
boxes, scores, idxs = [torch.from_numpy(g[k]).cuda() for k in ('rboxes','rscores','ridxs')]
by_score = torch.argsort(scores, descending=True, stable=True)
order = by_score[torch.argsort(idxs[by_score], stable=True)]
bs = boxes[order].contiguous(); cs = idxs[order].to(torch.int64).contiguous()
k = bs.shape[0]; W = (k+63)//64
lib = _lib.lib()
ws = torch.zeros(k*W, dtype=torch.int64, device='cuda'); keep = torch.zeros(k, dtype=torch.uint8, device='cuda')
rc = lib.sph2pob_nms_f32(G.ptr(bs), G.ptr(cs), ctypes.c_int64(k), 4, 1, ctypes.c_float(0.5), G.ptr(ws), G.ptr(keep), None)
torch.cuda.synchronize(); print('rc', rc)
mask = ws.cpu().numpy().view(np.uint64).reshape(k, W)
iou = S.sph2pob_efficient_iou(bs, bs).cpu().numpy(); c = cs.cpu().numpy()
exp = np.zeros((k, W), np.uint64)
for i in range(k):
    for j in range(i+1, k):
        if c[j]==c[i] and iou[i,j] > 0.5: exp[i, j//64] |= np.uint64(1) << np.uint64(j%64)
print('mask mismatches', (mask != exp).sum())
bad = np.argwhere(mask != exp)[:5]; print(bad, [(hex(mask[a,b]), hex(exp[a,b])) for a,b in bad])
# numpy sweep on exp
removed = np.zeros(k, bool); kp = np.zeros(k, bool)
for i in range(k):
    if not removed[i]:
        kp[i]=True
        for w in range(W):
            bits = int(exp[i,w])
            for b in range(64):
                if bits>>b & 1: removed[w*64+b]=True
print('keep mismatches vs numpy sweep', (kp != keep.cpu().numpy().astype(bool)).sum(), np.nonzero(kp != keep.cpu().numpy().astype(bool))[0][:10])
