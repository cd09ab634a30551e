"""world_size-2 data-parallel test on CPU (gloo): the N>1 path of trainer.py -- the flat one-all-reduce gradient
average (GradSync, default) and the torch DDP wrapper (SRK_USE_TORCH_DDP=1): identical replicas, and equality with a
single process at 2x batch."""
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
    os.environ["SRK_USE_TORCH_DDP"] = "1" if use_ddp else "0"
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
@pytest.mark.parametrize("use_ddp", [False, True], ids=["flat_allreduce", "torch_ddp"])
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
