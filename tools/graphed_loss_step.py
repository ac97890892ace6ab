"""GPU box: configs[2] as a user would run it when the host must not be in the way — the whole training step
(`Sph2PobIoULoss(mode='ciou')` forward + backward to the predictions) captured into hipGraphs and replayed: (a) the whole
step in one graph (torch's whole-network capture recipe), (b) `torch.cuda.make_graphed_callables`.  The launchers only enqueue and nothing is cached from inside a capture
(INTEGRATION.md §2), so the loss module needs no change.  Prints the per-step time of the eager step and of the graphed
one, and checks that the graphed loss / gradient equal the eager ones bit for bit."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import sph_retina_amd as S  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
g = torch.Generator(device='cpu').manual_seed(0)
u = torch.rand((n, 5), generator=g)
tgt = torch.stack([u[:, 0] * 360, 20 + u[:, 1] * 140, 5 + u[:, 2] * 60, 5 + u[:, 3] * 60, u[:, 4] * 180 - 90], 1).cuda()
pred = (tgt + torch.randn(n, 5, generator=g).cuda() * 2).clamp(min=1).requires_grad_(True)
loss_fn = S.Sph2PobIoULoss(mode='ciou')


def timeit(fn, reps=300):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def eager():
    pred.grad = None
    loss_fn(pred, tgt).backward()


eager()
torch.cuda.synchronize()
ref_loss = loss_fn(pred, tgt).detach().clone()
ref_grad = pred.grad.clone()
t_eager = timeit(eager)

# (a) the whole step — forward, backward, accumulation into pred.grad — captured once and replayed: torch's whole-network
# capture recipe (warm up on a side stream, drop the gradient, capture); a replay is one graph launch, no autograd on the host
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(3):
        eager()
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
pred.grad = None
whole = torch.cuda.CUDAGraph()
with torch.cuda.graph(whole):
    static_loss = loss_fn(pred, tgt)
    static_loss.backward()
pred.grad.zero_()
whole.replay()
torch.cuda.synchronize()
assert torch.equal(static_loss.detach(), ref_loss) and torch.equal(pred.grad, ref_grad)
t_whole = timeit(whole.replay)
print(f'pairs {n}: whole step replayed from one hipGraph {t_whole * 1e6:.1f} us ({n / t_whole:.3e} pairs/s), equal to the eager step bit for bit')

# (b) torch.cuda.make_graphed_callables (separate forward / backward graphs behind an autograd node of torch's own)
graphed = torch.cuda.make_graphed_callables(loss_fn, (pred, tgt))


def graphed_step():
    pred.grad = None
    graphed(pred, tgt).backward()


graphed_step()
torch.cuda.synchronize()
out = graphed(pred, tgt)
assert torch.equal(out.detach(), ref_loss), (float(out), float(ref_loss))
assert torch.equal(pred.grad, ref_grad)
t_graph = timeit(graphed_step)
print(f'pairs {n}: eager step {t_eager * 1e6:.1f} us ({n / t_eager:.3e} pairs/s), graphed step {t_graph * 1e6:.1f} us '
      f'({n / t_graph:.3e} pairs/s); loss and gradient equal the eager ones bit for bit')
