"""world_size-2 data-parallel test on CPU (gloo): the N>1 path of trainer.py -- the bucketed gradient average whose
all-reduces are launched from post-accumulate-grad hooks during backward (GradSync, default; one bucket and several) and
the torch DDP wrapper (SRK_USE_TORCH_DDP=1): identical replicas, and equality with a single process at 2x batch.
Also the hook-less form bench.py's hipGraph step uses (pack / reduce)."""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out, use_ddp):
    sys.path.insert(0, ROOT)
    os.environ["SRK_USE_TORCH_DDP"] = "1" if use_ddp == "torch_ddp" else "0"
    if use_ddp == "buckets":
        os.environ["SRK_BUCKET_BYTES"] = "4096"         # SRCNN: several buckets, all-reduces launched from the grad hooks
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(1)
    import sr_amd
    from sr_amd import trainer as T
    torch.manual_seed(0)
    m = sr_amd.SRCNN(scale_factor=2, optimizer="SGD")
    tr = T.Trainer(device="cpu", max_steps=3)
    full = [T.synthetic_batch(4, 3, 16, 2, 100 + s, "cpu") for s in range(3)]
    shard = [{"lr": b["lr"][rank * 2:(rank + 1) * 2], "hr": b["hr"][rank * 2:(rank + 1) * 2], "path": b["path"][rank * 2:(rank + 1) * 2]} for b in full]
    tr.fit(m, shard)
    torch.save({k: v.clone() for k, v in m.state_dict().items()}, os.path.join(out, f"r{rank}.pt"))
    torch.distributed.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("use_ddp", ["flat", "buckets", "torch_ddp"], ids=["one_bucket", "overlapped_buckets", "torch_ddp"])
def test_two_rank_ddp_equals_single_process(tmp_path, use_ddp):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path), use_ddp), nprocs=2, join=True)
    a, b = torch.load(tmp_path / "r0.pt"), torch.load(tmp_path / "r1.pt")
    for k in a:
        assert torch.equal(a[k], b[k]), f"replicas diverged at {k}"
    # single process, global batch 4 (mean loss == mean of the two shard means)
    for v in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        os.environ.pop(v, None)
    import sr_amd
    from sr_amd import trainer as T
    torch.manual_seed(0)
    m = sr_amd.SRCNN(scale_factor=2, optimizer="SGD")
    tr = T.Trainer(device="cpu", max_steps=3)
    tr.fit(m, [T.synthetic_batch(4, 3, 16, 2, 100 + s, "cpu") for s in range(3)])
    for k, v in m.state_dict().items():
        assert torch.allclose(v, a[k], rtol=1e-5, atol=1e-7), k


def _worker_pack_reduce(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(1)
    import sr_amd
    from sr_amd import trainer as T
    T.init_distributed("cpu")
    torch.manual_seed(0)
    m = sr_amd.SRCNN(scale_factor=2, optimizer="SGD")
    gs = T.GradSync(m, overlap=False, bucket_bytes=4096)
    gs.broadcast()
    b = T.synthetic_batch(4, 3, 16, 2, 7, "cpu")
    sh = {"lr": b["lr"][rank * 2:(rank + 1) * 2], "hr": b["hr"][rank * 2:(rank + 1) * 2]}
    m._calculate_losses(img_sr=m(sh["lr"]), img_hr=sh["hr"])["loss"].backward()
    gs.pack()
    gs.reduce()
    assert all(p.grad.data_ptr() == gs.views[p].data_ptr() for p in gs.params)
    torch.save([p.grad.clone() for p in m.parameters()], os.path.join(out, f"g{rank}.pt"))
    if rank == 0:
        torch.manual_seed(0)
        ref = sr_amd.SRCNN(scale_factor=2, optimizer="SGD")
        ref._calculate_losses(img_sr=ref(b["lr"]), img_hr=b["hr"])["loss"].backward()
        torch.save([p.grad.clone() for p in ref.parameters()], os.path.join(out, "ref.pt"))
    torch.distributed.destroy_process_group()


@pytest.mark.timeout(300)
def test_pack_reduce_form_matches_full_batch_gradient(tmp_path):
    mp.spawn(_worker_pack_reduce, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    g0, g1, ref = torch.load(tmp_path / "g0.pt"), torch.load(tmp_path / "g1.pt"), torch.load(tmp_path / "ref.pt")
    for a, b, r in zip(g0, g1, ref):
        assert torch.equal(a, b)
        assert float((a - r).abs().max()) <= 1e-4 * float(r.abs().max()) + 1e-12


def _worker_segments(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(1)
    import sr_amd
    from sr_amd import trainer as T
    T.init_distributed("cpu")
    torch.manual_seed(0)
    m = sr_amd.SRCNN(scale_factor=2, optimizer="SGD")
    for t in list(m.parameters()):
        torch.distributed.broadcast(t.data, src=0)
    opt = m.configure_optimizers()[0]
    ogs = T.OverlappedGraphStep(m, opt, segments=3, bucket_bytes=4096)
    full = [T.synthetic_batch(4, 3, 16, 2, 100 + s, "cpu") for s in range(3)]
    for i, b in enumerate(full):
        sh = {"lr": b["lr"][rank * 2:(rank + 1) * 2], "hr": b["hr"][rank * 2:(rank + 1) * 2]}
        if i == 0:
            ogs.prepare(sh)                       # the step that finds the segments' parameter groups
            assert len(ogs.groups) == 4 and [len(g) for g in ogs.groups] == [2, 2, 2, 0], [len(g) for g in ogs.groups]
        else:
            ogs.eager_step(sh)                    # segment by segment, the bucket all-reduces launched in between
    torch.save({k: v.clone() for k, v in m.state_dict().items()}, os.path.join(out, f"seg_r{rank}.pt"))
    torch.distributed.destroy_process_group()


@pytest.mark.timeout(300)
def test_segmented_backward_two_ranks_equals_single_process(tmp_path):
    """trainer.OverlappedGraphStep's launch-by-launch form on 2 gloo ranks (the backward pass of SRCNN cut into three autograd
    passes at its `ops.cut` points, each followed by the asynchronous all-reduces of the buckets it completed): identical replicas,
    equal to one process at twice the batch."""
    port = _free_port()
    mp.spawn(_worker_segments, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    a, b = torch.load(tmp_path / "seg_r0.pt"), torch.load(tmp_path / "seg_r1.pt")
    for k in a:
        assert torch.equal(a[k], b[k]), f"replicas diverged at {k}"
    for v in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        os.environ.pop(v, None)
    import sr_amd
    from sr_amd import trainer as T
    torch.manual_seed(0)
    m = sr_amd.SRCNN(scale_factor=2, optimizer="SGD")
    tr = T.Trainer(device="cpu", max_steps=3)
    tr.fit(m, [T.synthetic_batch(4, 3, 16, 2, 100 + s, "cpu") for s in range(3)])
    for k, v in m.state_dict().items():
        assert torch.allclose(v, a[k], rtol=1e-5, atol=1e-7), k


def _worker_shifted(rank, world, port, out):
    """The multi-rank hipGraph step's update ORDER on CPU: `GraphedStep.shifted_eager_step` is one replay of the optimizer-first
    graph launch by launch -- step k's update is applied at the start of step k + 1, `finish()` applies the last one -- with the
    learning rate changed between two steps the way a scheduler does (after `optimizer.step()` in the reference's order)."""
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(1)
    import sr_amd
    from sr_amd import trainer as T
    T.init_distributed("cpu")
    torch.manual_seed(0)
    m = sr_amd.SRCNN(scale_factor=2, optimizer="SGD")
    opt = m.configure_optimizers()[0]
    gsync = T.GradSync(m, overlap=False, bucket_bytes=4096)
    gsync.broadcast()
    gs = T.GraphedStep(m, m, opt, gsync, warm_steps=0)
    full = [T.synthetic_batch(4, 3, 16, 2, 200 + s, "cpu") for s in range(5)]
    for k, b in enumerate(full):
        sh = {"lr": b["lr"][rank * 2:(rank + 1) * 2], "hr": b["hr"][rank * 2:(rank + 1) * 2]}
        if k == 3:                                   # "scheduler.step()" after the third optimizer step
            for g in opt.param_groups:
                g["lr"] *= 0.1
        gs.shifted_eager_step(sh)
        assert gs.pending
    gs.finish()
    assert not gs.pending
    gsync.detach()
    torch.save({k: v.clone() for k, v in m.state_dict().items()}, os.path.join(out, f"s{rank}.pt"))
    torch.distributed.destroy_process_group()


@pytest.mark.timeout(300)
def test_shifted_update_order_two_ranks_equals_single_process(tmp_path):
    port = _free_port()
    mp.spawn(_worker_shifted, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    a, b = torch.load(tmp_path / "s0.pt"), torch.load(tmp_path / "s1.pt")
    for k in a:
        assert torch.equal(a[k], b[k]), f"replicas diverged at {k}"
    for v in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        os.environ.pop(v, None)
    import sr_amd
    from sr_amd import trainer as T
    torch.manual_seed(0)
    m = sr_amd.SRCNN(scale_factor=2, optimizer="SGD")
    opt = m.configure_optimizers()[0]
    for k in range(5):
        bt = T.synthetic_batch(4, 3, 16, 2, 200 + k, "cpu")
        if k == 3:
            for g in opt.param_groups:
                g["lr"] *= 0.1
        opt.zero_grad()
        m._calculate_losses(img_sr=m(bt["lr"]), img_hr=bt["hr"])["loss"].backward()
        opt.step()
    for k, v in m.state_dict().items():
        assert torch.allclose(v, a[k], rtol=1e-5, atol=1e-7), k


def test_pending_update_uses_the_hyper_parameters_of_its_gradients():
    """after an lr change the pending update (step k's) must use the OLD lr; flush() restores the new one."""
    sys.path.insert(0, ROOT)
    from sr_amd import trainer as T
    w = torch.nn.Parameter(torch.ones(4))
    opt = torch.optim.SGD([w], lr=0.5)
    gs = T.GraphedStep(torch.nn.Module(), None, opt, None)
    w.grad = torch.full((4,), 2.0)
    gs._mark_pending()
    opt.param_groups[0]["lr"] = 0.01
    gs.flush()
    assert torch.allclose(w.detach(), torch.full((4,), 1.0 - 0.5 * 2.0))
    assert opt.param_groups[0]["lr"] == 0.01 and not gs.pending
    gs.flush()                                         # nothing pending: no second update
    assert torch.allclose(w.detach(), torch.full((4,), 0.0))
