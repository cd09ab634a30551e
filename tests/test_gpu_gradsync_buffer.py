"""The multi-rank step structure on one GPU; GradSync's flat gradient buffer."""


import numpy as np


import pytest


import torch


from oracle import functional as OF, train as OT


pytestmark = pytest.mark.gpu


PREC = {torch.float16: 16, torch.bfloat16: "bf16"}


@pytest.fixture(scope="module")
def A():
    import sr_amd
    assert torch.cuda.is_available()
    sr_amd._lib.load()
    return sr_amd


RCAN_KW = dict(n_feats=64, n_resgroups=2, n_resblocks=3, reduction=16, scale_factor=2)


def test_bench_runs_the_multi_rank_step_structure_on_one_gpu():
    """`SRK_FORCE_DDP=1 python bench.py --batch 16`: a 1-rank `nccl` (RCCL) group makes bench.py take the code every rank of
    `--gpus N` takes -- GradSync buckets with the gradients written straight into the flat buffer, forward + backward as one
    hipGraph, the bucket all-reduce and the optimizer step behind it -- and the line must say so and train (loss falls)."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, SRK_FORCE_DDP="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29731", HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--batch", "16", "--steps", "30", "--warmup", "30", "--no-cpu-baseline",
                          "--no-roofline", "--no-other-configs", "--sustain-seconds", "0"], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads(out.stdout.strip().splitlines()[-1])
    assert d["config"]["hip_graph"] == "segmented" and d["config"]["grad_sync"] == "bucketed_allreduce", d["config"]
    assert d["value"] > 1000 and d["config"]["loss_after_timed_steps"] < d["config"]["loss_after_warmup"]


def test_gradsync_gradients_land_in_the_flat_buffer(A):
    """trainer.GradSync names each parameter's slice of its flat gradient buffer as the weight-gradient kernels' target: after a
    backward pass the conv gradients ARE views of the buffer (nothing for pack() to copy) and equal the gradients of a run
    without GradSync."""
    from sr_amd import trainer as T
    torch.manual_seed(0)
    m = A.EDSR(n_feats=64, n_resblocks=2, res_scale=0.1, scale_factor=2, precision="bf16").cuda()
    b = T.synthetic_batch(4, 3, 24, 2, 11, "cuda")
    m.training_step(b, 0)["loss"].backward()
    ref = {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None}
    for p in m.parameters():
        p.grad = None
    gs = T.GradSync(m, overlap=False)
    m.training_step(b, 0)["loss"].backward()
    torch.cuda.synchronize()
    inplace = sum(1 for p in gs.params if p.grad is not None and p.grad.data_ptr() == gs.views[p].data_ptr())
    assert inplace >= len(gs.params) - 2, f"{inplace} of {len(gs.params)} gradients were written in place"
    gs.pack()
    for k, p in m.named_parameters():
        if k in ref:
            assert torch.equal(p.grad, ref[k]), k
    gs.detach()
    assert all("_srk_grad_target" not in p.__dict__ for p in m.parameters())
