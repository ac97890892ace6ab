"""GPU box: hipGraph capture of a step that ends in `loss.backward()` INTO A LEAF's .grad (round-1 VERDICT #6).

    python tools/graph_capture_backward.py recipe     torch's documented whole-step capture recipe  -> must work
    python tools/graph_capture_backward.py stale      the pattern that crashed in round 1           -> torch / HIP limitation

What crashed in round 1 (gpurun_out/gd2a.log): the leaf's `.grad` already existed — allocated by an eager backward on
the default stream — when the capture began.  autograd's AccumulateGrad node runs on the stream the leaf's gradient was
first produced on, not on the capturing stream (torch prints "The AccumulateGrad node's stream does not match the stream
of the node that produced the incoming gradient ... may break CUDA graph capture"); the accumulation is then enqueued on
a stream outside the capture, which invalidates the capture, and on ROCm 7.2 `hipStreamEndCapture` dereferences the
invalidated graph instead of returning hipErrorStreamCaptureInvalidated: a segmentation fault inside
torch/cuda/graphs.py capture_end.  None of the library's launchers is involved (they only enqueue on the stream they are
given; the same step captured with torch.autograd.grad, which has no AccumulateGrad node, replays bit-identically).
The recipe that works is torch's own: warm up fwd + bwd on a side stream, drop the grads (`.grad = None`) so that
the captured backward ALLOCATES them from the graph's pool on the capturing stream, capture, replay.
"""
import faulthandler
import os
import sys

faulthandler.enable()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import sph_retina_amd as S  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else 'recipe'
n = 20000
g = torch.Generator(device='cpu')
g.manual_seed(0)
u = torch.rand((n, 4), generator=g)
anchors = torch.stack([u[:, 0] * 360, 20 + u[:, 1] * 140, 5 + u[:, 2] * 60, 5 + u[:, 3] * 60], 1).cuda()
target = (anchors + torch.randn(n, 4, generator=g).cuda() * 2).clamp(min=1)
deltas = (torch.randn(n, 4, generator=g).cuda() * 0.1).requires_grad_(True)
coder = S.DeltaXYWHSphBBoxCoder(target_stds=(0.1, 0.1, 0.2, 0.2))
loss_fn = S.Sph2PobIoULoss(mode='ciou')


def step():
    loss = loss_fn(coder.decode(anchors, deltas), target)
    loss.backward()
    return loss.detach()


if mode == 'stale':
    step()                      # eager backward on the default stream: deltas.grad now exists
    torch.cuda.synchronize()
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(3):          # warm-up on a side stream (lazy init, workspace allocation)
        if mode == 'recipe':
            deltas.grad = None
        step()
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
if mode == 'recipe':
    deltas.grad = None          # the captured backward allocates the gradient from the graph's pool
eager_grad = None
graph = torch.cuda.CUDAGraph()
print('capture begin', flush=True)
with torch.cuda.graph(graph):
    loss = step()
print('capture end', flush=True)
deltas.grad.zero_()
graph.replay()
torch.cuda.synchronize()
g1 = deltas.grad.clone()
ref, = torch.autograd.grad(loss_fn(coder.decode(anchors, deltas), target), deltas)
assert torch.equal(g1, ref), float((g1 - ref).abs().max())
assert torch.isfinite(loss)
print('replay ok: captured backward into the leaf equals the eager gradient', flush=True)
