"""hipGraphs with static descriptor tables replay like eager passes."""


import os


import sys


import pytest


import torch


ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


sys.path.insert(0, ROOT)


pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def A():
    import sr_amd
    assert torch.cuda.is_available()
    sr_amd._lib.load()
    return sr_amd


def _params(seed, shapes):
    g = torch.Generator().manual_seed(seed)
    return [torch.nn.Parameter((torch.rand(*s, generator=g) - 0.5).cuda()) for s in shapes]


_WG1X1_SCRIPT = r"""
import sys, torch
sys.path.insert(0, {root!r})
import sr_amd as A
torch.manual_seed(5)
out = {{}}
for tag, (n, h, w, cin, cout) in {{"head": (7, 48, 48, 32, 64), "ragged": (3, 13, 9, 32, 64), "one": (1, 1, 1, 32, 64), "sq": (2, 24, 24, 64, 64)}}.items():
    for dt in (torch.bfloat16, torch.float16):
        x = (torch.rand(n, h, w, cin, device="cuda") - 0.5).to(dt)
        x[..., 27:] = 0
        dy = (torch.rand(n, h, w, cout, device="cuda") - 0.5).to(dt)
        creal = 27 if cin == 32 else cin
        gw, gb = A.ops.wgrad_raw(x, dy, N=n, H=h, W=w, Cin=cin, Cout=cout, k=1, w_shape=(cout, creal, 1, 1), want_bias=True)
        ref = torch.einsum("nhwi,nhwo->oi", x.double(), dy.double())[:, :creal]
        out[f"{{tag}}_{{dt}}"] = (gw.cpu(), gb.cpu(), ref.cpu(), dy.double().sum((0, 1, 2)).cpu())
torch.save(out, sys.argv[1])
"""


@pytest.mark.parametrize("name,kw", [("EDSR", dict(n_feats=64, n_resblocks=3, res_scale=0.1)), ("WDSR", dict(type="B", n_feats=32, n_resblocks=2))])
def test_graph_with_static_descriptor_tables_replays_like_eager(A, name, kw):
    """ops.graph_capture: the grouped launches' descriptor tables are written once at capture time instead of by upload launches inside the
    graph.  Replays on NEW data must give the gradients an eager pass gives on that data (the tables hold addresses only), also after other
    work has run between the replays (the tables' memory stays the graph's)."""
    torch.manual_seed(7)
    m = getattr(A.models, name)(scale_factor=2, channels=3, precision="bf16", patch_size=48, **kw).cuda()
    params = [p for p in m.parameters() if p.requires_grad]
    x = torch.rand(4, 3, 24, 24, device="cuda")
    y = torch.rand(4, 3, 48, 48, device="cuda")

    def fwd_bwd():
        for p in params:
            p.grad = None
        loss = A.ops.l1_loss(m(x), y)
        A.ops.backward(loss)
        return loss

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            fwd_bwd()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with A.ops.graph_capture(g, stream=side) as holder:
        loss = fwd_bwd()
    assert holder.tables, "the backward pass of this model queues grouped launches: their tables must have gone through the holder"
    static_grads = [p.grad for p in params]
    for trial in range(3):
        x.copy_(torch.rand_like(x)); y.copy_(torch.rand_like(y))
        junk = [torch.rand(1 << 20, device="cuda") for _ in range(4)]         # other allocations and launches between the replays
        del junk
        g.replay()
        torch.cuda.synchronize()
        got = [t.clone() for t in static_grads]
        lg = loss.item()
        le = fwd_bwd().item()
        assert abs(lg - le) <= 1e-6 * max(1.0, abs(le))
        for a_, p in zip(got, params):
            assert torch.equal(a_, p.grad), (trial, tuple(a_.shape))
